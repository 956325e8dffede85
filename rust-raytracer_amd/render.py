"""Host-side mirror of the reference's Renderer (src/render.rs:150-265) on top of the C ABI.

`Renderer.render(buffer, mode)` and `Renderer.preview(mode)` keep the reference's
call shape: the frame buffer is a [height*width, 3] float32 array, row 0 = top
image row, passes add up and the caller divides by the pass count
(src/main.rs:262,288).  The work happens in librtcore.so on the GPU.
"""
import ctypes as C
import math

import numpy as np

from . import _abi
from ._abi import RT_MODE_COLORS, RT_MODE_IMAGE, RT_MODE_NORMALS  # noqa: F401

T_MIN = 0.00015  # src/render.rs:15
T_MAX = 100000.0  # src/render.rs:16


class Mode:
    """src/render.rs:31-36"""

    Normals = RT_MODE_NORMALS
    Colors = RT_MODE_COLORS
    Image = RT_MODE_IMAGE


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Scene:
    """An immutable world on the current HIP device (the reference's Arc<Bvh> plus environment and camera)."""

    def __init__(self, description, builder=_abi.RT_BUILD_DEFAULT):
        """builder: RT_BUILD_EXACT (the reference's tree, built on the host), RT_BUILD_DEVICE (built on the GPU) or
        RT_BUILD_DEFAULT (the environment's RTCORE_BUILD decides, exact otherwise)."""
        self._lib = _abi.load()
        self._desc_owner = description
        desc = description.to_desc()
        handle = C.c_void_p()
        _abi.check(self._lib.rt_scene_create_with(C.byref(desc), int(builder), C.byref(handle)))
        self._h = handle

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = _abi.RtSceneInfo()
        _abi.check(self._lib.rt_scene_info(self._h, C.byref(i)))
        return i

    def stats(self):
        """Counters since the last call (synchronises the device)."""
        s = _abi.RtStats()
        _abi.check(self._lib.rt_scene_stats(self._h, C.byref(s)))
        return s

    def set_stage_timing(self, enabled):
        _abi.check(self._lib.rt_set_stage_timing(self._h, int(bool(enabled))))

    def set_counting(self, enabled):
        _abi.check(self._lib.rt_set_counting(self._h, int(bool(enabled))))

    # ---- <Bvh as Hittable>::hit on a batch --------------------------------
    def closest_hit(self, org_dir, t_min=T_MIN, t_max=T_MAX):
        """org_dir: [n, 6] float32 (origin, direction).  Returns (prim_id[n], t[n], normal[n,3], point[n,3])."""
        rays = np.ascontiguousarray(org_dir, dtype=np.float32).reshape(-1, 6)
        n = len(rays)
        ids = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        nrm = np.empty((n, 3), dtype=np.float32)
        pt = np.empty((n, 3), dtype=np.float32)
        _abi.check(self._lib.rt_closest_hit_batch(self._h, _ptr(rays), t_min, t_max, n, _ptr(ids), _ptr(t), _ptr(nrm), _ptr(pt)))
        return ids, t, nrm, pt

    def closest_hit_device(self, d_org_dir, n, d_ids, d_t, d_normal, d_point, stream=0, t_min=T_MIN, t_max=T_MAX,
                           counted=False):
        """Device pointers (ints) in, nothing copied, no host synchronisation."""
        fn = self._lib.rt_closest_hit_batch_counted_device if counted else self._lib.rt_closest_hit_batch_device
        _abi.check(fn(self._h, d_org_dir, t_min, t_max, n, d_ids, d_t, d_normal, d_point, stream))

    def closest_hit_noalloc_device(self, d_org_dir, n, d_ids, d_t, d_normal, d_point, stream=0, t_min=T_MIN, t_max=T_MAX):
        """The same batch through the entry point that allocates nothing (at most RT_NOALLOC_MAX_RAYS rays)."""
        _abi.check(self._lib.rt_closest_hit_batch_noalloc_device(self._h, d_org_dir, t_min, t_max, n, d_ids, d_t, d_normal,
                                                                 d_point, stream))

    # ---- passes ------------------------------------------------------------
    @staticmethod
    def params(width, height, samples, max_bounce, light_clamp=math.inf, mode=Mode.Image, pass_index=0, seed=0,
               batch_paths=0, first_pixel=0, pixel_count=0):
        """pixel_count > 0 renders only pixels [first_pixel, first_pixel + pixel_count) of the frame (row-major from
        the top row); the buffer is the whole frame either way."""
        p = _abi.RtRenderParams()
        p.first_pixel, p.pixel_count = first_pixel, pixel_count
        p.width, p.height, p.samples, p.max_bounce = width, height, samples, max_bounce
        p.light_clamp = light_clamp
        p.mode, p.pass_index, p.seed, p.batch_paths = mode, pass_index, seed, batch_paths
        return p

    def render_pass(self, accum, params):
        """accum: float32 [height*width*3] host array, updated in place."""
        assert accum.dtype == np.float32 and accum.flags["C_CONTIGUOUS"]
        assert accum.size == params.width * params.height * 3
        _abi.check(self._lib.rt_render_pass(self._h, C.byref(params), _ptr(accum)))

    def render_pass_device(self, d_accum, params, stream=0):
        _abi.check(self._lib.rt_render_pass_device(self._h, C.byref(params), d_accum, stream))

    def preview(self, out, params):
        assert out.dtype == np.float32 and out.size == params.width * params.height * 3
        _abi.check(self._lib.rt_preview(self._h, C.byref(params), _ptr(out)))

    def preview_device(self, d_out, params, stream=0):
        _abi.check(self._lib.rt_preview_device(self._h, C.byref(params), d_out, stream))


class MultiScene:
    """Replicas of one scene on several GPUs of this process (rt_multi_gpu_*): passes are dealt round-robin, the
    device-resident accumulators are summed once (RCCL over xGMI, or peer copies) and added to the caller's frame."""

    def __init__(self, description, devices):
        self._lib = _abi.load()
        self._owner = description
        desc = description.to_desc()
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _abi.check(self._lib.rt_multi_gpu_init(C.byref(desc), devs, len(devices), C.byref(h)))
        self._h = h
        self.devices = list(devices)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_multi_gpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_passes(self, accum, params, first_pass, n_passes):
        """accum: float32 host array [height*width*3]; the sum of the passes is added to it."""
        assert accum.dtype == np.float32 and accum.flags["C_CONTIGUOUS"] and accum.size == params.width * params.height * 3
        _abi.check(self._lib.rt_render_passes_multi(self._h, C.byref(params), first_pass, n_passes, _ptr(accum)))

    def reduce(self, device_pointers, n_floats, how=0):
        """device_pointers[g] (ints) on devices[g]; the sum lands in device_pointers[0].  how: 0 auto, 1 RCCL, 2 peer copies."""
        arr = (C.c_void_p * len(device_pointers))(*device_pointers)
        _abi.check(self._lib.rt_reduce(self._h, arr, n_floats, how))

    def last_reduce(self):
        return {0: "none", 1: "rccl", 2: "peer"}[self._lib.rt_multi_gpu_last_reduce(self._h)]

    def stats(self):
        s = _abi.RtStats()
        _abi.check(self._lib.rt_multi_gpu_stats(self._h, C.byref(s)))
        return s


class Renderer:
    """Renderer { width, height, camera, world, sample_rate, max_bounce, hdr, light_clamp } (src/render.rs:151-160).

    camera, world and hdr are held by `scene`.  `seed` replaces the reference's unseeded generators;
    the pass counter advances with every render() call like the loop in src/main.rs:256-282.
    """

    def __init__(self, scene, width, height, sample_rate=128, max_bounce=8, light_clamp=math.inf, seed=0):
        self.scene = scene
        self.width, self.height = width, height
        self.sample_rate, self.max_bounce, self.light_clamp = sample_rate, max_bounce, light_clamp
        self.seed = seed
        self.passes_done = 0

    def render(self, buffer, mode=Mode.Image):
        """Returns buffer + one pass, as Renderer::render does (src/render.rs:162)."""
        out = np.array(buffer, dtype=np.float32, copy=True).reshape(-1)
        p = Scene.params(self.width, self.height, self.sample_rate, self.max_bounce, self.light_clamp, mode,
                         self.passes_done, self.seed)
        self.scene.render_pass(out, p)
        self.passes_done += 1
        return out.reshape(self.height * self.width, 3)

    def preview(self, mode=Mode.Image):
        """Renderer::preview (src/render.rs:222)"""
        out = np.zeros(self.width * self.height * 3, dtype=np.float32)
        p = Scene.params(self.width, self.height, 1, self.max_bounce, math.inf, mode, self.passes_done, self.seed)
        self.scene.preview(out, p)
        return out.reshape(self.height * self.width, 3)


def to_rgb(color, inv_gamma):
    """to_rgb (src/render.rs:39-45): packed 0xFFRRGGBB after ACES + gamma."""
    c = (C.c_float * 3)(*[float(x) for x in color])
    return int(_abi.load().rt_to_rgb(c, float(inv_gamma)))


def tonemap_device(d_rgb, n_pixels, divisor, inv_gamma, d_argb, stream=0):
    """to_rgb(color / divisor, inv_gamma) for a whole frame on the device (src/render.rs:39-55, src/main.rs:266-271)."""
    _abi.check(_abi.load().rt_tonemap_device(d_rgb, n_pixels, divisor, inv_gamma, d_argb, stream))


def bilateral_filter_device(d_in, d_out, width, height, diameter, sigma_i, sigma_s, stream=0):
    """One iteration of bilateral_filter on the device (src/filter.rs:14-48)."""
    _abi.check(_abi.load().rt_bilateral_filter_device(d_in, d_out, width, height, diameter, sigma_i, sigma_s, stream))
