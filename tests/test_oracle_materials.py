"""Known answers for Material::scatter (src/material.rs:28-136,176-183), derived by hand from the formulas, checked
on the oracle.  The device scatter is covered by the bit-identical render-pass tests (it must reproduce every
scattered ray of the oracle to pass them)."""
import ctypes as C
import math

import numpy as np
import pytest

AIR = 1.00028


def _scatter(oracle, pkg, mat, d, n, key, bounce=0, t=1.0):
    A = pkg._abi
    m = A.RtMaterial(mat["type"], A.F3(*mat["albedo"]), mat.get("p0", 0.0), mat.get("p1", 0.0))
    d = np.asarray(d, np.float32)
    d = d / np.linalg.norm(d)
    od = np.concatenate([np.array([0, 0, 0], np.float32) - d, d]).astype(np.float32)
    point = np.zeros(3, np.float32)
    nn = np.asarray(n, np.float32)
    att = np.zeros(3, np.float32)
    ray = np.zeros(6, np.float32)
    oracle.lib().oracle_scatter(C.byref(m), od.ctypes.data, C.c_float(t), point.ctypes.data, nn.ctypes.data, key, bounce,
                                att.ctypes.data, ray.ctypes.data)
    return att, ray[:3], ray[3:]


def _schlick(c, ni, nt):
    r0 = ((ni - nt) / (ni + nt)) ** 2
    return min(max(r0 + (1 - r0) * (1 - c) ** 5, 0.0), 1.0)


def _keys(oracle, n=64):
    return [oracle.lib().oracle_path_key(5, i, 0) for i in range(n)]


def _draw(oracle, key_index, dim, bounce=0):
    return oracle.lib().oracle_draw(5, key_index, 0, bounce, dim)


def test_metal_mirror(pkg, oracle):
    A = pkg._abi
    mat = {"type": A.RT_MAT_METAL, "albedo": (0.8, 0.6, 0.2), "p0": 0.0}
    for i, key in enumerate(_keys(oracle, 8)):
        att, o, d = _scatter(oracle, pkg, mat, (0, -1, 0), (0, 1, 0), key)
        assert np.allclose(d, (0, 1, 0), atol=1e-6) and np.allclose(o, 0)
        assert np.allclose(att, mat["albedo"], atol=1e-6)  # cos = 1: no Fresnel lift
        att, o, d = _scatter(oracle, pkg, mat, (1, -1, 0), (0, 1, 0), key)
        assert np.allclose(d, np.array([1, 1, 0]) / math.sqrt(2), atol=1e-6)
        lift = (1 - math.sqrt(0.5)) ** 5
        assert np.allclose(att, np.array(mat["albedo"]) + (1 - np.array(mat["albedo"])) * lift, atol=1e-6)
    # grazing: (1 - cos)^5 -> 1, attenuation -> 1, never above (clamp)
    att, _, _ = _scatter(oracle, pkg, mat, (1, -1e-4, 0), (0, 1, 0), 1)
    assert np.all(att <= 1.0) and np.all(att > 0.99)


def test_metal_rough_normal_stays_in_the_jitter_ball(pkg, oracle):
    A = pkg._abi
    mat = {"type": A.RT_MAT_METAL, "albedo": (1, 1, 1), "p0": 0.3}
    for key in _keys(oracle, 64):
        _, _, d = _scatter(oracle, pkg, mat, (0, -1, 0), (0, 1, 0), key)
        # reflect(d, jn) with jn within 0.3 of n: the reflected ray is within 2 * asin(0.3) of the mirror direction
        assert d[1] > math.cos(2 * math.asin(0.3)) - 1e-5
        assert abs(np.linalg.norm(d) - 1) < 1e-5


def test_lambertian(pkg, oracle):
    A = pkg._abi
    mat = {"type": A.RT_MAT_LAMBERTIAN, "albedo": (0.3, 0.5, 0.7)}
    n = np.array([0.0, 0.6, 0.8], np.float32)
    cos = []
    for i, key in enumerate(_keys(oracle, 256)):
        att, o, d = _scatter(oracle, pkg, mat, (0, -1, 0), n, key)
        assert np.array_equal(att, np.array(mat["albedo"], np.float32))
        c = float(np.dot(d, n))
        # z = sqrt(1 - r2) along the normal (src/random.rs:25-32, src/ray.rs:46-50)
        r2 = _draw(oracle, i, 1)
        assert abs(c - math.sqrt(1 - r2)) < 1e-5
        cos.append(c)
    assert abs(np.mean(cos) - 2 / 3) < 0.04  # E[cos] of a cosine-weighted hemisphere
    # light: albedo > 1 passes through unchanged (the caller treats it as an emitter, src/render.rs:66,73-77)
    att, _, _ = _scatter(oracle, pkg, {"type": A.RT_MAT_LAMBERTIAN, "albedo": (15, 15, 15)}, (0, -1, 0), n, 3)
    assert np.array_equal(att, np.float32([15, 15, 15]))


def test_glossy_branches(pkg, oracle):
    A = pkg._abi
    mat = {"type": A.RT_MAT_GLOSSY, "albedo": (0.2, 0.4, 0.6), "p0": 1.0, "p1": 0.0}
    p_normal = _schlick(1.0, AIR, 2.0)  # reflectance 1 -> nt = 2
    assert abs(p_normal - 0.11093) < 2e-4
    n_spec = 0
    for i, key in enumerate(_keys(oracle, 256)):
        att, _, d = _scatter(oracle, pkg, mat, (0, -1, 0), (0, 1, 0), key)
        choice = _draw(oracle, i, 5)
        if choice <= p_normal - 1e-6:
            assert np.allclose(att, 0.9) and np.allclose(d, (0, 1, 0), atol=1e-6)
            n_spec += 1
        elif choice > p_normal + 1e-6:
            assert np.allclose(att, mat["albedo"]) and d[1] >= 0
    assert 10 <= n_spec <= 50  # about 11 % of 256


def test_dielectric_enter_exit_and_total_internal_reflection(pkg, oracle):
    A = pkg._abi
    mat = {"type": A.RT_MAT_DIELECTRIC, "albedo": (0.1, 0.2, 0.3), "p0": 1.52, "p1": 0.0}
    p_in = _schlick(1.0, AIR, 1.52)
    assert abs(p_in - 0.0425247) < 1e-6
    seen = set()
    for i, key in enumerate(_keys(oracle, 256)):
        choice = _draw(oracle, i, 5)
        if abs(choice - p_in) < 1e-6:
            continue
        # entering along the normal: 0.9 tint; refraction keeps the direction, reflection reverses it
        att, _, d = _scatter(oracle, pkg, mat, (0, -1, 0), (0, 1, 0), key)
        assert np.allclose(att, 0.9)
        assert np.allclose(d, (0, 1, 0) if choice <= p_in else (0, -1, 0), atol=1e-6)
        seen.add(choice <= p_in)
        # entering at 45 degrees: Snell
        att, _, d = _scatter(oracle, pkg, mat, (1, -1, 0), (0, 1, 0), key)
        c = math.sqrt(0.5)
        if choice > _schlick(c, AIR, 1.52) + 1e-6:
            sin_t = (AIR / 1.52) * math.sqrt(0.5)
            assert np.allclose(d, (sin_t, -math.sqrt(1 - sin_t * sin_t), 0), atol=2e-6)
        # leaving along the normal after t = 2 inside: Beer-Lambert exp(-2 t absorption)
        att, _, d = _scatter(oracle, pkg, mat, (0, 1, 0), (0, 1, 0), key, t=2.0)
        assert np.allclose(att, np.exp(-4.0 * np.array(mat["albedo"])), rtol=2e-6)
        p_out = _schlick(1.0, 1.52, AIR)
        if abs(choice - p_out) > 1e-6:
            assert np.allclose(d, (0, -1, 0) if choice <= p_out else (0, 1, 0), atol=1e-6)
        # leaving at a grazing angle: (ni / nt) sin > 1 -> always the mirror direction about -n
        att, _, d = _scatter(oracle, pkg, mat, (1, 0.1, 0), (0, 1, 0), key)
        e = np.array([1, -0.1, 0]) / math.sqrt(1.01)
        assert np.allclose(d, e, atol=1e-6)
    assert seen == {True, False}


def test_dielectric_critical_angle(pkg, oracle):
    A = pkg._abi
    mat = {"type": A.RT_MAT_DIELECTRIC, "albedo": (0, 0, 0), "p0": 1.52, "p1": 0.0}
    crit = math.asin(AIR / 1.52)
    for delta, total in ((-0.01, False), (0.01, True)):
        a = crit + delta
        d_in = (math.sin(a), math.cos(a), 0)  # leaving: d . n > 0
        outs = set()
        for key in _keys(oracle, 64):
            _, _, d = _scatter(oracle, pkg, mat, d_in, (0, 1, 0), key)
            outs.add(bool(d[1] > 0))  # refracted rays keep going up, reflected ones turn down
        assert outs == ({False} if total else {True, False}) or (not total and outs == {True})
