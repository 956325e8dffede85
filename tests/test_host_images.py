"""The C++ host's image files (host/image_io.hpp through libhostio.so, no GPU): the Radiance .hdr reader that stands
in for the radiant crate (src/io.rs:106-111), its writer, and the PNG writer (src/io.rs:310-336)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def rgbe_quantise(rgb):
    """What an RGBE pixel can hold of float RGB, written independently of the C++ code: shared exponent of the
    largest component, 8-bit truncated mantissas, decoded as mantissa * 2^(e - 136)."""
    rgb = np.asarray(rgb, np.float32).reshape(-1, 3)
    m = rgb.max(axis=1)
    mant, e = np.frexp(m)
    scale = np.where(m > 1e-32, mant.astype(np.float32) * np.float32(256.0) / np.where(m > 1e-32, m, 1), 0).astype(np.float32)
    q = np.floor(rgb * scale[:, None]).astype(np.int64)
    out = (q * np.exp2((e - 8).astype(np.float64))[:, None]).astype(np.float32)
    out[m <= 1e-32] = 0
    return out


def test_hdr_round_trip_flat_and_rle(pkg, tmp_path):
    rng = np.random.default_rng(5)
    img = (rng.random((37, 61, 3)) ** 4 * 200).astype(np.float32)
    img[3, 5:40] = (0.5, 0.25, 0.125)  # a long run
    img[10, :, :] = 0.0  # black pixels have exponent byte 0
    flat, rle = tmp_path / "a.hdr", tmp_path / "b.hdr"
    pkg.hostio.save_hdr(flat, img)
    pkg.hostio.save_hdr(rle, img, rle=True)
    a, b = pkg.hostio.load_hdr(flat), pkg.hostio.load_hdr(rle)
    assert a.shape == (37, 61, 3)
    assert np.array_equal(a, b)
    assert np.array_equal(a.reshape(-1, 3), rgbe_quantise(img))
    assert os.path.getsize(flat) == len(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 37 +X 61\n") + 37 * 61 * 4


def test_hdr_hand_made_scanlines(pkg, tmp_path):
    head = b"#?RADIANCE\n# a comment\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n"
    # one run-length encoded scanline of 8 pixels: R = a run of 8, G = 8 literals, B = a run of 4 then 4 literals,
    # E = a run of 8 exponents 129 (a pixel's value is mantissa * 2^(129 - 136))
    scan = bytes([2, 2, 0, 8]) + bytes([128 + 8, 128]) + bytes([8, 1, 2, 3, 4, 5, 6, 7, 8]) + bytes([128 + 4, 10, 4, 1, 2, 3, 4]) + bytes([128 + 8, 129])
    p = tmp_path / "rle.hdr"
    p.write_bytes(head + b"-Y 1 +X 8\n" + scan)
    img = pkg.hostio.load_hdr(p)
    want = np.zeros((1, 8, 3), np.float32)
    want[0, :, 0] = 128
    want[0, :, 1] = np.arange(1, 9)
    want[0, :, 2] = [10, 10, 10, 10, 1, 2, 3, 4]
    assert np.array_equal(img, want / np.float32(128.0))
    # narrow images are stored flat (a run-length scanline needs at least 8 pixels); row 0 of the file is row 0 of the image
    px = bytes([64, 32, 16, 128, 0, 0, 0, 0, 255, 255, 255, 136]) + bytes([1, 2, 3, 129, 4, 5, 6, 130, 7, 8, 9, 131])
    p = tmp_path / "flat.hdr"
    p.write_bytes(head + b"-Y 2 +X 3\n" + px)
    img = pkg.hostio.load_hdr(p)
    assert img.shape == (2, 3, 3)
    assert np.array_equal(img[0], np.array([[64, 32, 16], [0, 0, 0], [255, 255, 255]], np.float32) * np.float32([[2.0 ** -8], [0], [1.0]]))
    assert np.array_equal(img[1, 2], np.float32([7, 8, 9]) * np.float32(2.0 ** -5))
    # other orientations, a missing magic line and a truncated file are refused
    for name, data in (("flip.hdr", head + b"+Y 2 +X 3\n" + px), ("nomagic.hdr", b"FORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 3\n" + px),
                       ("short.hdr", head + b"-Y 2 +X 3\n" + px[:13])):
        q = tmp_path / name
        q.write_bytes(data)
        with pytest.raises(RuntimeError):
            pkg.hostio.load_hdr(q)


def test_committed_sky_fixtures(pkg):
    flat = pkg.hostio.load_hdr(os.path.join(GOLDEN, "sky_512x256_flat.hdr"))
    rle = pkg.hostio.load_hdr(os.path.join(GOLDEN, "sky_512x256_rle.hdr"))
    assert flat.shape == (256, 512, 3) and np.array_equal(flat, rle)
    assert np.array_equal(flat.reshape(-1, 3), rgbe_quantise(pkg.scenes.sky_map(512, 256)))
    assert 50.0 < flat.max() < 52.0 and flat.min() > 0.1  # the sun disk over the gradient
    # the tea-set scene takes its environment through this reader
    env = pkg.scenes.teaset(env_size=(512, 256)).env
    assert np.array_equal(env, flat)
    big = pkg.scenes.sky_environment(1024, 512)
    assert np.array_equal(big.reshape(-1, 3), rgbe_quantise(pkg.scenes.sky_map(1024, 512)))


def test_png_writer(pkg, tmp_path):
    from pngdec import decode_png

    rng = np.random.default_rng(9)
    W, H = 97, 53
    noise = rng.integers(0, 1 << 24, W * H, dtype=np.uint32) | np.uint32(0xFF000000)
    x, y = np.meshgrid(np.arange(W, dtype=np.uint32), np.arange(H, dtype=np.uint32))
    smooth = (np.uint32(0xFF000000) | ((2 * x) << 16) | ((4 * y) << 8) | (x + y)).reshape(-1).astype(np.uint32)
    for name, img in (("noise", noise), ("smooth", smooth)):
        p = tmp_path / (name + ".png")
        pkg.hostio.save_png(p, img, W, H)
        px, filters, size, idat = decode_png(p)
        assert size == (W, H) and np.array_equal(px, img)
        if name == "smooth":
            assert (filters != 0).any()  # a row filter was chosen: gradients are cheaper as differences
            assert idat < 0.2 * W * H * 3  # and zlib level 9 did its work
