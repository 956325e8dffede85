"""ctypes mirror of include/rtcore.h and loader of librtcore.so.

The library is the product: there is no Python or CPU fallback.  `load()` raises
if the shared object is missing, and every compute entry point of the library
returns RT_ERR_DEVICE when no HIP device is visible.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTCORE_LIB selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("RTCORE_LIB") or os.path.join(_HERE, "librtcore.so")

RT_OK = 0
RT_ERR_INVALID_ARGUMENT = -1
RT_ERR_OUT_OF_MEMORY = -2
RT_ERR_DEVICE = -3
RT_ERR_EMPTY_SCENE = -4
RT_ERR_INTERNAL = -5

RT_MAT_DIELECTRIC, RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_GLOSSY = 0, 1, 2, 3
RT_OBJ_SPHERE, RT_OBJ_ABOX, RT_OBJ_CUBE, RT_OBJ_TRIANGLE, RT_OBJ_MESH = 0, 1, 2, 3, 4
RT_MODE_NORMALS, RT_MODE_COLORS, RT_MODE_IMAGE = 0, 1, 2
RT_NOALLOC_MAX_RAYS = 262144
RT_BUILD_DEFAULT, RT_BUILD_EXACT, RT_BUILD_DEVICE = 0, 1, 2

F3 = C.c_float * 3
F9 = C.c_float * 9


class RtMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("albedo", F3), ("p0", C.c_float), ("p1", C.c_float)]


class RtObject(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("material", C.c_int32),
        ("a", F3),
        ("b", F3),
        ("c", F3),
        ("tri_vertices", F9),
        ("tri_normals", F9),
        ("flag", C.c_int32),
        ("mesh", C.c_int32),
    ]


class RtMesh(C.Structure):
    _fields_ = [
        ("positions", C.POINTER(C.c_float)),
        ("normals", C.POINTER(C.c_float)),
        ("indices", C.POINTER(C.c_uint32)),
        ("n_vertices", C.c_uint64),
        ("n_faces", C.c_uint64),
        ("translation", F3),
        ("scale", F3),
        ("rotation", F3),
        ("cull_backface", C.c_int32),
        ("reserved", C.c_int32),
    ]


class RtCamera(C.Structure):
    _fields_ = [
        ("eye", F3),
        ("lookat", F3),
        ("vup", F3),
        ("fov_degrees", C.c_float),
        ("aperture", C.c_float),
        ("focus_dist", C.c_float),
    ]


class RtEnvMap(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_float)), ("width", C.c_uint32), ("height", C.c_uint32)]


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("materials", C.POINTER(RtMaterial)),
        ("n_materials", C.c_uint64),
        ("objects", C.POINTER(RtObject)),
        ("n_objects", C.c_uint64),
        ("meshes", C.POINTER(RtMesh)),
        ("n_meshes", C.c_uint64),
        ("camera", RtCamera),
        ("env", RtEnvMap),
    ]


class RtRenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("samples", C.c_uint32),
        ("max_bounce", C.c_uint32),
        ("light_clamp", C.c_float),
        ("mode", C.c_int32),
        ("pass_index", C.c_uint32),
        ("seed", C.c_uint32),
        ("batch_paths", C.c_uint64),
        ("first_pixel", C.c_uint64),
        ("pixel_count", C.c_uint64),
    ]


class RtSceneInfo(C.Structure):
    _fields_ = [
        ("n_primitives", C.c_uint64),
        ("n_triangles", C.c_uint64),
        ("n_wide_nodes", C.c_uint64),
        ("n_binary_nodes", C.c_uint64),
        ("wide_depth", C.c_uint32),
        ("node_bytes", C.c_uint32),
        ("primitive_bytes", C.c_uint32),
        ("builder", C.c_uint32),
        ("device_bytes", C.c_uint64),
        ("bounds_min", F3),
        ("bounds_max", F3),
        ("build_ms", C.c_double),
    ]


class RtStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("paths", C.c_uint64),
        ("node_visits", C.c_uint64),
        ("prim_tests", C.c_uint64),
        ("extend_ms", C.c_double),
        ("shade_ms", C.c_double),
        ("other_ms", C.c_double),
        ("extend_launches", C.c_uint64),
        ("node_steps", C.c_uint64),
        ("node_step_lanes", C.c_uint64),
        ("leaf_steps", C.c_uint64),
        ("leaf_step_lanes", C.c_uint64),
        ("l1_lookups", C.c_uint64),
    ]


# every symbol include/rtcore.h declares: name -> (restype, argtypes)
_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int32)
_VP = C.c_void_p
SYMBOLS = {
    "rt_init": (C.c_int, [C.c_int]),
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
    "rt_scene_create": (C.c_int, [C.POINTER(RtSceneDesc), C.POINTER(_VP)]),
    "rt_scene_create_with": (C.c_int, [C.POINTER(RtSceneDesc), C.c_int, C.POINTER(_VP)]),
    "rt_scene_create_replicas_with": (
        C.c_int,
        [C.POINTER(RtSceneDesc), C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(_VP)],
    ),
    "rt_scene_validate": (C.c_int, [C.POINTER(RtSceneDesc), C.POINTER(RtSceneInfo)]),
    "rt_scene_destroy": (C.c_int, [_VP]),
    "rt_scene_info": (C.c_int, [_VP, C.POINTER(RtSceneInfo)]),
    "rt_scene_stats": (C.c_int, [_VP, C.POINTER(RtStats)]),
    "rt_render_pass": (C.c_int, [_VP, C.POINTER(RtRenderParams), _VP]),
    "rt_render_pass_device": (C.c_int, [_VP, C.POINTER(RtRenderParams), _VP, _VP]),
    "rt_preview": (C.c_int, [_VP, C.POINTER(RtRenderParams), _VP]),
    "rt_preview_device": (C.c_int, [_VP, C.POINTER(RtRenderParams), _VP, _VP]),
    "rt_closest_hit_batch": (C.c_int, [_VP, _VP, C.c_float, C.c_float, C.c_size_t, _VP, _VP, _VP, _VP]),
    "rt_closest_hit_batch_device": (C.c_int, [_VP, _VP, C.c_float, C.c_float, C.c_size_t, _VP, _VP, _VP, _VP, _VP]),
    "rt_closest_hit_batch_counted_device": (
        C.c_int,
        [_VP, _VP, C.c_float, C.c_float, C.c_size_t, _VP, _VP, _VP, _VP, _VP],
    ),
    "rt_closest_hit_batch_noalloc_device": (
        C.c_int,
        [_VP, _VP, C.c_float, C.c_float, C.c_size_t, _VP, _VP, _VP, _VP, _VP],
    ),
    "rt_scene_create_replicas": (C.c_int, [C.POINTER(RtSceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(_VP)]),
    "rt_multi_gpu_init": (C.c_int, [C.POINTER(RtSceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(_VP)]),
    "rt_multi_gpu_destroy": (C.c_int, [_VP]),
    "rt_multi_gpu_count": (C.c_int, [_VP]),
    "rt_multi_gpu_scene": (_VP, [_VP, C.c_int]),
    "rt_render_passes_multi": (C.c_int, [_VP, C.POINTER(RtRenderParams), C.c_uint32, C.c_uint32, _VP]),
    "rt_reduce": (C.c_int, [_VP, C.POINTER(_VP), C.c_size_t, C.c_int]),
    "rt_multi_gpu_last_reduce": (C.c_int, [_VP]),
    "rt_multi_gpu_stats": (C.c_int, [_VP, C.POINTER(RtStats)]),
    "rt_comm_available": (C.c_int, []),
    "rt_comm_unique_id": (C.c_int, [_VP]),
    "rt_comm_init_rank": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(_VP)]),
    "rt_comm_reduce": (C.c_int, [_VP, _VP, C.c_size_t, C.c_int, _VP]),
    "rt_comm_destroy": (C.c_int, [_VP]),
    "rt_set_stage_timing": (C.c_int, [_VP, C.c_int]),
    "rt_set_counting": (C.c_int, [_VP, C.c_int]),
    "rt_tonemap_device": (C.c_int, [_VP, C.c_size_t, C.c_float, C.c_float, _VP, _VP]),
    "rt_bilateral_filter_device": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, _VP]),
    "rt_tonemap": (C.c_int, [_VP, C.c_size_t, C.c_float, C.c_float, _VP]),
    "rt_bilateral_filter": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float]),
    "rt_to_rgb": (C.c_uint32, [_FP, C.c_float]),
}

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rtcore error %d: %s" % (code, message))
        self.code = code


def load():
    """Loads librtcore.so and sets the prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "librtcore.so is not built (%s): run `make -C rust-raytracer_amd` or __graft_entry__.build()" % LIB_PATH
        )
    # One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64 and ask for it by file name, while
    # librtcore asks for the soname: a process that loads librtcore first gets the system runtime, a later
    # `import torch` brings a second one, and torch then finds no device.  With torch imported first the loader
    # binds librtcore to the runtime torch brought, whose streams and allocations the callers hand in anyway.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != RT_OK:
        msg = load().rt_last_error()
        raise RtError(rc, msg.decode() if msg else "")
