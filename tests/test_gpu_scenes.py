"""GPU tests at the configurations' shapes: the tea-set meshes, image statistics against the oracle with a
different seed, the closest-hit microbenchmark's checks, and the C++ host end to end."""
import math
import os
import subprocess
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTRENDER = os.path.join(ROOT, "rust-raytracer_amd", "rtrender")


def test_teaset_hits_and_pass(pkg, oracle):
    d = pkg.scenes.teaset(env_size=(512, 256))
    scene, ora = pkg.Scene(d), oracle.OracleScene(d)
    mn, mx = ora.bounds()
    rays = np.concatenate([pkg.scenes.random_rays(200_000, mn, mx, seed=3, inflate=1.2),
                           pkg.scenes.random_rays(200_000, (-2.5, 0, -1), (2, 2, 2.5), seed=4, inflate=1.0)])
    ids, t, nrm, pt = scene.closest_hit(rays)
    oids, ot, onrm, opt = ora.closest_hit(rays)
    assert np.array_equal(ids, oids)
    assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    assert np.array_equal(nrm.view(np.uint32), onrm.view(np.uint32))
    assert (ids > 0).sum() > 50_000  # mesh triangles, both culled and two-sided models
    W, H = 128, 72
    p = pkg.Scene.params(W, H, 8, 8, seed=21)
    acc, oacc = np.zeros(W * H * 3, np.float32), np.zeros(W * H * 3, np.float32)
    scene.render_pass(acc, p)
    ora.render_pass(oacc, p)
    assert np.array_equal(acc.view(np.uint32), oacc.view(np.uint32))


def _luminance(a):
    a = a.reshape(-1, 3)
    return 0.2126 * a[:, 0] + 0.7152 * a[:, 1] + 0.0722 * a[:, 2]


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("which", ["rtiow", "cornell", "teaset"])
def test_converged_4096spp_image_statistics_against_oracle_with_other_seeds(pkg, oracle, which):
    """The acceptance criterion stated for the unseeded reference, at its stated 4096 samples per pixel (64 passes of
    64): the GPU render with seed A against oracle renders with seeds B and C.
      mean luminance: within 1 % of each oracle render's;
      per-pixel RMSE: at or below the oracle's own seed-to-seed noise floor, RMSE(B, C), with no slack factor.
    The image the GPU render is held against is the oracle's best estimate of the converged frame, the mean of its
    B and C renders.  (Against B or C alone the comparison is a coin flip by construction: A, B and C are three
    independent draws of one estimator, so RMSE(A, B) and RMSE(B, C) have the same expectation, 2 sigma^2; against
    the mean of two the expectation is 1.5 sigma^2, so a correct renderer passes with a margin of about 13 % and a
    biased one does not.  With equal seeds the images are bit-identical, tested elsewhere.)"""
    if which == "rtiow":
        d, W, H, clamp = pkg.scenes.rtiow(seed=1), 160, 90, math.inf
    elif which == "cornell":
        d, W, H, clamp = pkg.scenes.cornell(), 160, 90, math.inf
    else:
        d, W, H, clamp = pkg.scenes.teaset(env_size=(512, 256)), 128, 72, math.inf
    scene, ora = pkg.Scene(d), oracle.OracleScene(d)
    spp, passes = 64, 64

    def render(backend, seed):
        acc = np.zeros(W * H * 3, np.float32)
        for i in range(passes):
            backend.render_pass(acc, pkg.Scene.params(W, H, spp, 8, light_clamp=clamp, pass_index=i, seed=seed))
        return acc / passes

    gpu = render(scene, 101)
    o1, o2 = render(ora, 202), render(ora, 303)
    lg, l1, l2 = _luminance(gpu), _luminance(o1), _luminance(o2)
    assert abs(lg.mean() - l1.mean()) <= 0.01 * l1.mean() and abs(lg.mean() - l2.mean()) <= 0.01 * l2.mean()
    floor = math.sqrt(float(np.mean((l1 - l2) ** 2)))
    rmse = math.sqrt(float(np.mean((lg - 0.5 * (l1 + l2)) ** 2)))
    pairwise = [math.sqrt(float(np.mean((lg - l) ** 2))) for l in (l1, l2)]
    assert rmse <= floor, (which, rmse, floor, pairwise)


def test_closest_hit_microbench_checks(pkg, oracle):
    """The microbenchmark's two checks at reduced size: ids exact on an oracle-checked subsample, and an id
    checksum that is stable between runs and between chunkings of the same batch."""
    import torch

    d = pkg.scenes.mesh_scene(n_triangles=250_000, rings=40, segments=40, seed=7)
    scene, ora = pkg.Scene(d), oracle.OracleScene(d)
    info = scene.info()
    n = 3_000_000
    rays = pkg.scenes.random_rays(n, info.bounds_min[:], info.bounds_max[:], seed=0x5EED)
    d_rays = torch.from_numpy(rays).cuda()
    ids = torch.empty(n, dtype=torch.int32, device="cuda")
    t = torch.empty(n, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    sums = []
    for _ in range(2):
        scene.closest_hit_device(d_rays.data_ptr(), n, ids.data_ptr(), t.data_ptr(), None, None, stream)
        torch.cuda.synchronize()
        sums.append(zlib.crc32(ids.cpu().numpy().tobytes()))
    assert sums[0] == sums[1]
    whole = ids.cpu().numpy().copy()
    half = n // 2  # two calls on halves give the same ids as one call
    scene.closest_hit_device(d_rays.data_ptr(), half, ids.data_ptr(), t.data_ptr(), None, None, stream)
    scene.closest_hit_device(d_rays.data_ptr() + half * 24, n - half, ids.data_ptr() + half * 4, t.data_ptr() + half * 4,
                             None, None, stream)
    torch.cuda.synchronize()
    assert np.array_equal(whole, ids.cpu().numpy())
    sub = np.random.default_rng(1).choice(n, 300_000, replace=False)
    oids, ot, _, _ = ora.closest_hit(rays[sub])
    assert int((whole[sub] != oids).sum()) == 0
    assert np.array_equal(t.cpu().numpy()[sub].view(np.uint32), ot.view(np.uint32))
    st = scene.stats()
    assert st.rays == 3 * n  # two whole batches and one batch in two halves


def _decode_png(path, W, H):
    from pngdec import decode_png

    px, _, size, _ = decode_png(path)
    assert size == (W, H)
    return px


def test_rtrender_end_to_end(tmp_path, pkg, oracle):
    """The C++ host (reference command line) on a scene file written in the reference's schema gives the same
    frame as the Python mirror of Renderer::render with the same seed."""
    d = pkg.scenes.cornell()
    scene_file = tmp_path / "cornell.ron"
    scene_file.write_text(pkg.scene.to_ron(d))
    out = tmp_path / "out.png"
    W, H, spp, passes = 120, 80, 8, 3
    log = subprocess.check_output([RTRENDER, str(scene_file), "--width", str(W), "--height", str(H), "-s", str(spp), "-p",
                                   str(passes), "-b", "8", "--seed", "5", "--out", str(out)]).decode()
    assert "Frame took" in log and "Average frame time" in log and "finished pass 3" in log
    r = pkg.Renderer(pkg.Scene(d), W, H, sample_rate=spp, max_bounce=8, seed=5)
    buf = np.zeros((W * H, 3), np.float32)
    for _ in range(passes):
        buf = r.render(buf, pkg.Mode.Image)
    buf = buf / np.float32(passes)
    want = np.array([pkg.to_rgb(c, 1.0 / 2.2) for c in buf], dtype=np.uint32)
    assert np.array_equal(_decode_png(out, W, H), want)
    # -f: the three filter iterations of src/main.rs:114-132 before the display transform
    out_f = tmp_path / "filtered.png"
    subprocess.check_output([RTRENDER, str(scene_file), "--width", str(W), "--height", str(H), "-s", str(spp), "-p",
                             str(passes), "-b", "8", "--seed", "5", "-f", "--out", str(out_f)])
    cur = np.ascontiguousarray(buf, np.float32)
    for i in (1, 2, 3):
        nxt = np.empty_like(cur)
        oracle.lib().oracle_bilateral_filter(cur.ctypes.data, nxt.ctypes.data, W, H, 9 // i, 0.05 / i, 1.0 / i)
        cur = nxt
    want_f = np.array([pkg.to_rgb(c, 1.0 / 2.2) for c in cur], dtype=np.uint32)
    assert np.array_equal(_decode_png(out_f, W, H), want_f)


@pytest.mark.gpu
def test_answers_do_not_depend_on_the_wide_tree(pkg, oracle, monkeypatch):
    """The interior of the wide tree is free (which binary nodes root wide nodes, or a rebuilt interior): every
    variant must return the reference's primitive and distance for every ray."""
    experiments = b"+experiments" in pkg.load().rt_version()
    if not experiments:
        pytest.skip("the alternative tree shapes are compiled only into `make experiments` builds (load one through RTCORE_LIB)")
    d = pkg.scenes.mesh_scene(n_triangles=60_000, rings=32, segments=32, seed=11)
    ora = oracle.OracleScene(d)
    rays = pkg.scenes.random_rays(400_000, *ora.bounds(), seed=5)
    want_ids, want_t, _, _ = ora.closest_hit(rays)
    for name, value in (("RTCORE_COLLAPSE", "greedy"), ("RTCORE_INTERIOR", "sah"), (None, None)):
        if name:
            monkeypatch.setenv(name, value)
        scene = pkg.Scene(d)
        if name:
            monkeypatch.delenv(name)
        ids, t, _, _ = scene.closest_hit(rays)
        assert np.array_equal(ids, want_ids), name
        assert np.array_equal(t.view(np.uint32), want_t.view(np.uint32)), name
