"""Host-side scene build of the product (rt_scene_validate, no device) against the oracle's tree."""
import ctypes as C

import numpy as np
import pytest


def validate(pkg, S):
    info = pkg._abi.RtSceneInfo()
    desc = S.to_desc()
    pkg._abi.check(pkg.load().rt_scene_validate(C.byref(desc), C.byref(info)))
    return info


@pytest.mark.parametrize("name", ["cornell", "rtiow", "mesh", "single", "pair", "triple"])
def test_tree_matches_oracle(pkg, oracle, name):
    if name == "cornell":
        S = pkg.scenes.cornell()
    elif name == "rtiow":
        S = pkg.scenes.rtiow(seed=5)
    elif name == "mesh":
        S = pkg.scenes.mesh_scene(n_triangles=40_000, rings=16, segments=24, seed=3)
    else:
        S = pkg.SceneDescription()
        m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
        for i in range({"single": 1, "pair": 2, "triple": 3}[name]):
            S.add_sphere((2.0 * i, 0, 0), 0.5, m)
    info = validate(pkg, S)
    ora = oracle.OracleScene(S)
    assert info.n_primitives == ora.num_prims() == S.num_primitives()
    assert info.n_binary_nodes == ora.num_nodes()  # same reference-shaped tree
    mn, mx = ora.bounds()
    assert np.array_equal(np.array(info.bounds_min[:], np.float32), mn)
    assert np.array_equal(np.array(info.bounds_max[:], np.float32), mx)
    assert info.node_bytes == 64 and info.primitive_bytes == 48
    assert 1 <= info.n_wide_nodes <= max(1, info.n_binary_nodes)


def test_visit_order_is_a_permutation(pkg, oracle):
    S = pkg.scenes.mesh_scene(n_triangles=20_000, rings=16, segments=16, seed=1)
    order = oracle.OracleScene(S).visit_order()
    assert sorted(order.tolist()) == list(range(S.num_primitives()))


def test_large_build_is_threaded_and_deterministic(pkg):
    S = pkg.scenes.mesh_scene(n_triangles=300_000, rings=48, segments=48, seed=2)
    a, b = validate(pkg, S), validate(pkg, S)
    for f in ("n_wide_nodes", "n_binary_nodes", "wide_depth", "device_bytes"):
        assert getattr(a, f) == getattr(b, f)
