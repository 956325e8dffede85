"""The C-ABI library loads on a machine without a GPU, exports every symbol include/rtcore.h declares,
and refuses to compute without a device (there is no CPU path)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtcore.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(pkg):
    lib = C.CDLL(pkg._abi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "librtcore.so does not export " + n
    assert set(names) == set(pkg._abi.SYMBOLS), "Python prototypes out of sync with include/rtcore.h"


def test_struct_sizes_match_header(pkg, tmp_path):
    """Compiles include/rtcore.h as plain C (it must be valid C) and compares struct sizes with the ctypes mirror."""
    names = ["RtMaterial", "RtObject", "RtMesh", "RtCamera", "RtEnvMap", "RtSceneDesc", "RtRenderParams", "RtSceneInfo",
             "RtStats"]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "rtcore.h"\nint main(void){' +
                   "".join('printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in names) + "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for n in names:
        assert int(out[n]) == C.sizeof(getattr(pkg._abi, n)), n


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the behaviour without a device")
def test_no_device_is_a_loud_error(pkg):
    lib = pkg.load()
    d = pkg.scenes.cornell()
    with pytest.raises(pkg.RtError) as e:
        pkg.Scene(d)
    assert e.value.code == pkg._abi.RT_ERR_DEVICE
    assert b"no CPU path" in lib.rt_last_error()
    assert lib.rt_init(0) == pkg._abi.RT_ERR_DEVICE


def test_argument_errors(pkg):
    lib = pkg.load()
    info = pkg._abi.RtSceneInfo()
    S = pkg.SceneDescription()
    desc = S.to_desc()
    assert lib.rt_scene_validate(C.byref(desc), C.byref(info)) == pkg._abi.RT_ERR_EMPTY_SCENE
    S.add_sphere((0, 0, 0), 1.0, 5)  # material index out of range
    desc = S.to_desc()
    assert lib.rt_scene_validate(C.byref(desc), C.byref(info)) == pkg._abi.RT_ERR_INVALID_ARGUMENT
    assert b"material" in lib.rt_last_error()
    assert lib.rt_scene_validate(None, C.byref(info)) == pkg._abi.RT_ERR_INVALID_ARGUMENT
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
    S.add_mesh(np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32), np.array([[0, 1, 7]], np.uint32), m)
    desc = S.to_desc()
    assert lib.rt_scene_validate(C.byref(desc), C.byref(info)) == pkg._abi.RT_ERR_INVALID_ARGUMENT
    assert b"vertex index" in lib.rt_last_error()
    assert lib.rt_scene_destroy(None) == 0
