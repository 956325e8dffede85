"""Display-side kernels (tonemap + pack, bilateral filter) against the oracle's restatements."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tonemap_device_matches_to_rgb(pkg, oracle):
    import torch

    rng = np.random.default_rng(3)
    n = 200_000
    rgb = (rng.random((n, 3)) ** 3 * 40.0).astype(np.float32)
    rgb[:1000] *= 0.01
    rgb[1000:1010] = [[-1, 0.5, 2], [np.nan, 1, 1], [np.inf, 0, 0], [0, 0, 0], [1, 1, 1], [0.18, 0.18, 0.18], [2, 0.5, 0.1],
                      [1e-9, 1e-9, 1e-9], [1e9, 1e9, 1e9], [3, 3, 3]]
    divisor, gamma = 4.0, 1.0 / 2.2
    want = np.empty(n, np.uint32)
    oracle.lib().oracle_to_rgb_array(rgb.ctypes.data, n, divisor, gamma, want.ctypes.data)
    d_rgb = torch.from_numpy(rgb).cuda()
    d_out = torch.empty(n, dtype=torch.int32, device="cuda")
    pkg.render.tonemap_device(d_rgb.data_ptr(), n, divisor, gamma, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(np.uint32)
    # both sides take the power through double precision, so the packed pixels agree exactly
    finite = np.isfinite(rgb).all(axis=1)
    assert np.array_equal(got[finite], want[finite])
    assert np.all((got >> 24) == 255)
    # the known answers of the host function hold on the device too (divisor 4 folded in)
    assert got[1004] == want[1004] and got[1005] == want[1005] and got[1006] == want[1006]


def test_bilateral_filter_device(pkg, oracle):
    import torch

    rng = np.random.default_rng(5)
    W, H = 96, 64
    img = (rng.random((H * W, 3)) * 0.05 + 0.4).astype(np.float32)
    img[::97] += 0.5  # fireflies
    stream = torch.cuda.current_stream().cuda_stream
    cur = img.copy()
    d_a, d_b = torch.from_numpy(img).cuda(), torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
    for i in (1, 2, 3):  # the schedule of src/main.rs:114-132
        want = np.empty_like(cur)
        oracle.lib().oracle_bilateral_filter(cur.ctypes.data, want.ctypes.data, W, H, 9 // i, 0.05 / i, 1.0 / i)
        pkg.render.bilateral_filter_device(d_a.data_ptr(), d_b.data_ptr(), W, H, 9 // i, 0.05 / i, 1.0 / i, stream)
        torch.cuda.synchronize()
        got = d_b.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
        cur = want
        d_a.copy_(torch.from_numpy(want))
    # the left edge picks up the far edge (the reference's u32 wrap): column 0 differs from a clamped filter
    assert np.isfinite(cur).all()


def test_post_argument_errors(pkg):
    lib = pkg.load()
    assert lib.rt_tonemap_device(None, 10, 1.0, 1.0, None, None) == pkg._abi.RT_ERR_INVALID_ARGUMENT
    assert lib.rt_bilateral_filter_device(None, None, 4, 4, 3, 0.1, 1.0, None) == pkg._abi.RT_ERR_INVALID_ARGUMENT


@pytest.mark.gpu
def test_host_buffer_variants_equal_the_device_ones(pkg, oracle):
    lib = pkg.load()
    rng = np.random.default_rng(9)
    W, H = 64, 40
    img = (rng.random((H * W, 3)) * 0.3 + 0.2).astype(np.float32)
    out = np.empty_like(img)
    want = np.empty_like(img)
    pkg._abi.check(lib.rt_bilateral_filter(img.ctypes.data, out.ctypes.data, W, H, 9, 0.05, 1.0))
    oracle.lib().oracle_bilateral_filter(img.ctypes.data, want.ctypes.data, W, H, 9, 0.05, 1.0)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    packed = np.empty(H * W, np.uint32)
    wanted = np.empty(H * W, np.uint32)
    pkg._abi.check(lib.rt_tonemap(out.ctypes.data, H * W, 2.0, 1.0 / 2.2, packed.ctypes.data))
    oracle.lib().oracle_to_rgb_array(want.ctypes.data, H * W, 2.0, 1.0 / 2.2, wanted.ctypes.data)
    assert np.array_equal(packed, wanted)
    assert lib.rt_bilateral_filter(img.ctypes.data, img.ctypes.data, W, H, 9, 0.05, 1.0) == pkg._abi.RT_ERR_INVALID_ARGUMENT
