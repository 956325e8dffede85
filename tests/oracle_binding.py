"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("oracle/liboracle.so is not built: run `make -C oracle`")
        L = C.CDLL(LIB_PATH)
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_scene_create.restype = C.c_int
        L.oracle_scene_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_scene_num_prims.restype = C.c_int64
        L.oracle_scene_num_prims.argtypes = [C.c_void_p]
        L.oracle_scene_num_nodes.restype = C.c_int64
        L.oracle_scene_num_nodes.argtypes = [C.c_void_p]
        L.oracle_visit_order.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_scene_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_closest_hit_batch.restype = C.c_int
        L.oracle_closest_hit_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_size_t,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_render_pass.restype = C.c_int
        L.oracle_render_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_preview.restype = C.c_int
        L.oracle_preview.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_last_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_set_counting.argtypes = [C.c_int]
        L.oracle_schlick.restype = C.c_float
        L.oracle_schlick.argtypes = [C.c_float, C.c_float, C.c_float]
        L.oracle_sky_gradient.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_sky.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.oracle_warp.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.oracle_math.restype = C.c_float
        L.oracle_math.argtypes = [C.c_int, C.c_float, C.c_float]
        L.oracle_math_array.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_draw.restype = C.c_float
        L.oracle_draw.argtypes = [C.c_uint32] * 5
        L.oracle_pixel_jitter.argtypes = [C.c_uint32] * 4 + [C.c_void_p]
        L.oracle_gen_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_void_p]
        L.oracle_scatter.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_uint32,
                                     C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_path_key.restype = C.c_uint32
        L.oracle_path_key.argtypes = [C.c_uint32] * 3
        L.oracle_to_rgb.restype = C.c_uint32
        L.oracle_to_rgb.argtypes = [C.c_void_p, C.c_float]
        L.oracle_render_range.restype = C.c_int
        L.oracle_render_range.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        L.oracle_to_rgb_array.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p]
        L.oracle_bilateral_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_threads():
    return min(16, os.cpu_count() or 1)


class OracleScene:
    def __init__(self, description):
        self._owner = description
        desc = description.to_desc()
        h = C.c_void_p()
        rc = lib().oracle_scene_create(C.byref(desc), C.byref(h))
        if rc != 0:
            raise RuntimeError("oracle_scene_create failed: %d %s" % (rc, lib().oracle_last_error()))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_scene_destroy(self._h)
            self._h = None

    def num_prims(self):
        return int(lib().oracle_scene_num_prims(self._h))

    def num_nodes(self):
        return int(lib().oracle_scene_num_nodes(self._h))

    def visit_order(self):
        out = np.empty(self.num_prims(), dtype=np.int32)
        lib().oracle_visit_order(self._h, _p(out))
        return out

    def bounds(self):
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        lib().oracle_scene_bounds(self._h, _p(mn), _p(mx))
        return mn, mx

    def closest_hit(self, org_dir, t_min=0.00015, t_max=100000.0, threads=None):
        rays = np.ascontiguousarray(org_dir, dtype=np.float32).reshape(-1, 6)
        n = len(rays)
        ids = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        nrm = np.empty((n, 3), np.float32)
        pt = np.empty((n, 3), np.float32)
        rc = lib().oracle_closest_hit_batch(self._h, _p(rays), t_min, t_max, n, _p(ids), _p(t), _p(nrm), _p(pt),
                                            threads or default_threads())
        assert rc == 0
        return ids, t, nrm, pt

    def render_pass(self, accum, params, threads=None):
        rc = lib().oracle_render_pass(self._h, C.byref(params), _p(accum), threads or default_threads())
        assert rc == 0, rc

    def render_range(self, accum, params, first_pixel, n_pixels, threads=None):
        rc = lib().oracle_render_range(self._h, C.byref(params), first_pixel, n_pixels, _p(accum), threads or default_threads())
        assert rc == 0, rc

    def preview(self, out, params, threads=None):
        rc = lib().oracle_preview(self._h, C.byref(params), _p(out), threads or default_threads())
        assert rc == 0, rc

    def counters(self):
        c = np.zeros(3, np.uint64)
        lib().oracle_last_counters(self._h, _p(c))
        return dict(rays=int(c[0]), node_visits=int(c[1]), prim_tests=int(c[2]))
