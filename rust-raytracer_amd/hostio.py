"""ctypes binding of libhostio.so: the C++ host's Radiance .hdr reader / writer and PNG writer
(host/image_io.hpp), so that environments handed to the core in Python went through the same RGBE decoding the
headless host applies to a scene file's `hdr:` entry (src/io.rs:106-111)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libhostio.so")
        if not os.path.exists(path):
            raise RuntimeError("libhostio.so is not built: run `make -C rust-raytracer_amd`")
        L = C.CDLL(path)
        L.hostio_last_error.restype = C.c_char_p
        L.hostio_load_hdr.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
        L.hostio_save_hdr.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
        L.hostio_save_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("hostio: " + lib().hostio_last_error().decode())


def load_hdr(path):
    """[height, width, 3] float32, first scanline first (row 0 = "up" for the renderer, src/render.rs:95-96)."""
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib().hostio_load_hdr(os.fsencode(path), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 3), np.float32)
    _check(lib().hostio_load_hdr(os.fsencode(path), None, None, out.ctypes.data, out.size))
    return out


def save_hdr(path, rgb, rle=False):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    _check(lib().hostio_save_hdr(os.fsencode(path), w, h, rgb.ctypes.data, int(bool(rle))))


def save_png(path, argb, width, height):
    argb = np.ascontiguousarray(argb, np.uint32)
    assert argb.size == width * height
    _check(lib().hostio_save_png(os.fsencode(path), argb.ctypes.data, width, height))
