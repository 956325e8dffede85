"""The oracle against the hand-derived known answers of tests/golden/kat.json (CPU), and the
GPU core against the same answers (gpu-marked)."""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


def scene_of(pkg, objects):
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
    for o in objects:
        (kind, a), = o.items()
        if kind == "sphere":
            S.add_sphere(a["center"], a["radius"], m)
        elif kind == "abox":
            S.add_axis_box(a["center"], a["size"], m)
        elif kind == "cube":
            S.add_box(a["center"], a["size"], a["rotation"], m)
        else:
            S.add_triangle(a["vertices"], m, normals=a["normals"], two_sided=a["two_sided"])
    return S


def run_hit(backend, S, case):
    ray = np.array([case["ray"]], dtype=np.float32)
    return backend(S).closest_hit(ray, case.get("t_min", 0.00015), case.get("t_max", 100000.0))


def check_hit_case(ids, t, nrm, pt, case):
    if not case["hit"]:
        assert ids[0] == -1, case["name"]
        return
    assert ids[0] == 0, case["name"]
    if "t_tol" in case:  # rotated boxes: the rotor's sines and cosines are rounded
        assert abs(float(t[0]) - case["t"]) <= case["t_tol"], (case["name"], float(t[0]))
    else:
        assert t[0] == np.float32(case["t"]), case["name"]
    tol = case.get("tol", 1e-6)
    assert np.allclose(nrm[0], case["normal"], atol=tol), (case["name"], nrm[0])
    assert np.allclose(pt[0], case["point"], atol=tol), (case["name"], pt[0])


@pytest.mark.parametrize("case", KAT["hits"], ids=lambda c: c["name"])
def test_hit_known_answers_oracle(pkg, oracle, case):
    check_hit_case(*run_hit(oracle.OracleScene, scene_of(pkg, case.get("objects") or [case["object"]]), case), case)


@pytest.mark.parametrize("case", KAT["order"], ids=lambda c: c["name"])
def test_visit_order_known_answers_oracle(pkg, oracle, case):
    ids, _, _, _ = run_hit(oracle.OracleScene, scene_of(pkg, case["objects"]), case)
    assert ids[0] == case["prim_id"]


@pytest.mark.gpu
def test_known_answers_gpu(pkg):
    for case in KAT["hits"]:
        check_hit_case(*run_hit(pkg.Scene, scene_of(pkg, case.get("objects") or [case["object"]]), case), case)
    for case in KAT["order"]:
        ids, _, _, _ = run_hit(pkg.Scene, scene_of(pkg, case["objects"]), case)
        assert ids[0] == case["prim_id"], case["name"]


def test_schlick(oracle):
    for c in KAT["schlick"]:
        assert abs(oracle.lib().oracle_schlick(c["cosine"], c["ni"], c["nt"]) - c["value"]) < 2e-7, c


def test_sky_gradient(oracle):
    for c in KAT["sky_gradient"]:
        d = np.array(c["dir"], np.float32)
        out = np.zeros(3, np.float32)
        oracle.lib().oracle_sky_gradient(d.ctypes.data, out.ctypes.data)
        assert np.allclose(out, c["value"], atol=3e-7), (out, c)


def test_to_rgb_oracle_and_library(pkg, oracle):
    for c in KAT["to_rgb"]:
        col = np.array(c["color"], np.float32)
        want = (255 << 24) | (c["rgb"][0] << 16) | (c["rgb"][1] << 8) | c["rgb"][2]
        assert oracle.lib().oracle_to_rgb(col.ctypes.data, 1.0 / c["gamma"]) == want, c
        assert pkg.to_rgb(c["color"], 1.0 / c["gamma"]) == want, c  # host helper of librtcore, needs no device


def test_warps(oracle):
    kinds = {"disk": 0, "cosine": 1, "sphere": 2}
    n = np.zeros(3, np.float32)
    for c in KAT["warps"]:
        out = np.zeros(3, np.float32)
        oracle.lib().oracle_warp(kinds[c["kind"]], c["r"][0], c["r"][1], n.ctypes.data, out.ctypes.data)
        assert np.allclose(out, c["value"], atol=c["tol"]), (c, out)


def test_hemisphere_is_about_the_normal(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = rng.normal(size=3).astype(np.float32)
        r1, r2 = rng.random(2)
        out = np.zeros(3, np.float32)
        oracle.lib().oracle_warp(3, r1, r2, n.ctypes.data, out.ctypes.data)
        assert abs(np.linalg.norm(out) - 1) < 1e-5
        cos = float(out @ (n / np.linalg.norm(n)))
        assert abs(cos - math.sqrt(1 - r2)) < 1e-5  # z = sqrt(1 - r2), src/random.rs:26


def _closed_room(pkg, albedo):
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((albedo, albedo, albedo)))
    S.add_axis_box((0, 0, 0), (-10, -10, -10), m)
    S.set_camera((0, 0, 0), (0, 0, 1), 60.0, aperture=0.0, focus_dist=1.0)
    return S


def _bounce_loop_checks(pkg, backend):
    B = KAT["bounce_loop"]
    c = B["closed_room"]
    S = _closed_room(pkg, c["albedo"])
    W, H, spp = 64, 64, 64
    acc = np.zeros(W * H * 3, np.float32)
    backend(S).render_pass(acc, pkg.Scene.params(W, H, spp, c["depth"], seed=11))
    assert abs(float(acc.mean()) - c["expected_mean"]) < c["tol"], acc.mean()
    # depth 0: 1 * 0.01
    acc = np.zeros(W * H * 3, np.float32)
    backend(S).render_pass(acc, pkg.Scene.params(W, H, 2, 0, seed=11))
    assert np.all(acc == np.float32(B["depth_zero"]["expected"]))
    # directly viewed emitter, with and without clamp
    e = B["emitter"]
    S = _closed_room(pkg, e["albedo"])
    for clamp, want in ((math.inf, e["expected"][0]), (e["light_clamp"], e["expected"][1])):
        acc = np.zeros(W * H * 3, np.float32)
        backend(S).render_pass(acc, pkg.Scene.params(W, H, 2, 8, light_clamp=clamp, seed=11))
        assert np.all(acc == np.float32(want)), (clamp, acc.min(), acc.max())


def test_bounce_loop_closed_forms_oracle(pkg, oracle):
    _bounce_loop_checks(pkg, oracle.OracleScene)


@pytest.mark.gpu
def test_bounce_loop_closed_forms_gpu(pkg):
    _bounce_loop_checks(pkg, pkg.Scene)


def test_primary_miss_returns_sky_unscaled(pkg, oracle):
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
    S.add_sphere((0, 0, -100), 1.0, m)  # behind the camera
    S.set_camera((0, 0, 0), (0, 1, 0.0001), 0.05, aperture=0.0, focus_dist=1.0)
    W = H = 8
    acc = np.zeros(W * H * 3, np.float32)
    oracle.OracleScene(S).render_pass(acc, pkg.Scene.params(W, H, 1, 8, seed=1))
    want = KAT["sky_gradient"][0]["value"]
    assert np.allclose(acc.reshape(-1, 3), want, atol=1e-3)  # the narrow view looks straight up


def test_nan_scrub(pkg, oracle):
    """A zero-normal hit (box point matching no face) gives NaN radiance; the per-sample scrub keeps pixels finite."""
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
    # a box far from the origin: ray.at(t) lands more than 1e-4 off every face, so no face matches (src/tracer/cube.rs:58-65)
    S.add_axis_box((30000.0, 0.0, 0.0), (2.0, 20000.0, 20000.0), m)
    S.set_camera((0, 0, 0), (1, 0.3, 0.2), 40.0, aperture=0.0, focus_dist=1.0)
    W = H = 16
    acc = np.zeros(W * H * 3, np.float32)
    oracle.OracleScene(S).render_pass(acc, pkg.Scene.params(W, H, 4, 4, seed=3))
    assert np.isfinite(acc).all()
