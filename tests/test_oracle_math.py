"""The oracle's pinned elementary functions against libm (numpy float64), and its generators' distributions."""
import numpy as np


def _call(oracle, fn, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    yy = None if y is None else np.ascontiguousarray(y, np.float32)
    oracle.lib().oracle_math_array(fn, x.ctypes.data, None if yy is None else yy.ctypes.data, out.ctypes.data, x.size)
    return out


def _ulps(a, ref):
    ref32 = ref.astype(np.float32)
    spacing = np.spacing(np.maximum(np.abs(ref32), np.float32(1e-30)))
    return np.abs(a.astype(np.float64) - ref) / spacing


def test_sin_cos(oracle):
    x = np.concatenate([np.linspace(-7, 7, 200001), np.linspace(0, 2 * np.pi, 100001), [0.0, np.pi / 2, np.pi]]).astype(np.float32)
    for fn, ref in ((0, np.sin), (1, np.cos)):
        got = _call(oracle, fn, x)
        err = np.abs(got.astype(np.float64) - ref(x.astype(np.float64)))
        assert err.max() < 1.5e-7  # absolute: |sin|,|cos| <= 1
    assert np.isnan(_call(oracle, 0, np.array([np.nan, np.inf, 1e9], np.float32))).all()


def test_acos(oracle):
    x = np.linspace(-1, 1, 200001).astype(np.float32)
    got = _call(oracle, 2, x)
    assert _ulps(got, np.arccos(x.astype(np.float64))).max() <= 4
    assert got[0] == np.float32(np.pi) and got[-1] == 0.0
    assert np.isnan(_call(oracle, 2, np.array([1.0001, -1.5, np.nan], np.float32))).all()


def test_atan2(oracle):
    rng = np.random.default_rng(0)
    y, x = rng.normal(size=200000).astype(np.float32), rng.normal(size=200000).astype(np.float32)
    got = _call(oracle, 3, y, x)
    err = np.abs(got.astype(np.float64) - np.arctan2(y.astype(np.float64), x.astype(np.float64)))
    assert err.max() < 5e-7
    assert _call(oracle, 3, np.array([0.0], np.float32), np.array([-1.0], np.float32))[0] == np.float32(np.pi)


def test_exp(oracle):
    x = np.linspace(-80, 10, 200001).astype(np.float32)
    got = _call(oracle, 4, x)
    assert _ulps(got, np.exp(x.astype(np.float64))).max() <= 4
    assert _call(oracle, 4, np.array([0.0], np.float32))[0] == 1.0
    assert _call(oracle, 4, np.array([-100.0], np.float32))[0] == 0.0


def test_uniform_draws(oracle):
    L = oracle.lib()
    u = np.array([L.oracle_draw(7, p, s, b, d) for p in range(40) for s in range(8) for b in range(4) for d in range(7)])
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    assert np.all(u * 8388608 == np.round(u * 8388608))  # 23-bit multiples of 2^-23, like fastrand's f32()
    # different dimensions / bounces / samples are decorrelated
    a = np.array([L.oracle_draw(7, p, 0, 0, 0) for p in range(4000)])
    b = np.array([L.oracle_draw(7, p, 0, 0, 1) for p in range(4000)])
    c = np.array([L.oracle_draw(7, p, 1, 0, 0) for p in range(4000)])
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.05 and abs(np.corrcoef(a, c)[0, 1]) < 0.05


def test_pixel_jitter_is_an_r2_sequence(oracle):
    L = oracle.lib()
    out = np.zeros(2, np.float32)
    pts = []
    for s in range(256):
        L.oracle_pixel_jitter(3, 1234, 5, s, out.ctypes.data)
        pts.append(out.copy())
    pts = np.array(pts, np.float64)
    assert pts.min() >= 0 and pts.max() < 1
    g = 1.32471795724474602596
    d = (pts[1:] - pts[:-1]) % 1.0
    assert np.allclose(d[:, 0], 1 / g, atol=1e-6) and np.allclose(d[:, 1], 1 / g / g, atol=1e-6)
    # well stratified: every cell of a 8x8 grid gets 2..6 of 256 points
    h, _, _ = np.histogram2d(pts[:, 0], pts[:, 1], bins=8, range=[[0, 1], [0, 1]])
    assert h.min() >= 2 and h.max() <= 6
