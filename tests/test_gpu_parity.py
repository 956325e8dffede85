"""GPU parity: the HIP core through its C ABI against the CPU oracle on the same seeded inputs.

Bar: primitive ids exact; t, normal and point within 1e-5 relative as the acceptance
criterion, and in fact bit-identical, because device and oracle perform the same IEEE
operation sequence; rendered passes bit-identical for the same reason.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel_err(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.maximum(1e-30, np.abs(b).astype(np.float64)))) if a.size else 0.0


def check_hits(pkg, oracle, desc, rays, t_min=0.00015, t_max=100000.0, expect_hits=True):
    scene = pkg.Scene(desc)
    ora = oracle.OracleScene(desc)
    ids, t, nrm, pt = scene.closest_hit(rays, t_min, t_max)
    oids, ot, onrm, opt = ora.closest_hit(rays, t_min, t_max)
    mism = int((ids != oids).sum())
    assert mism == 0, "%d of %d primitive ids differ (first at ray %d: %d vs %d)" % (
        mism, len(ids), int(np.argmax(ids != oids)), ids[np.argmax(ids != oids)], oids[np.argmax(ids != oids)])
    hit = oids >= 0
    if expect_hits:
        assert hit.sum() > 0
    # tolerance stated by the acceptance criterion
    assert _rel_err(t[hit], ot[hit]) <= 1e-5
    assert np.max(np.abs(nrm[hit] - onrm[hit]), initial=0.0) <= 1e-5
    # what the design delivers: identical bits (NaN normals compare by bit pattern)
    assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    assert np.array_equal(nrm.view(np.uint32), onrm.view(np.uint32))
    assert np.array_equal(pt.view(np.uint32), opt.view(np.uint32))
    scene.close()
    return ids


def camera_like_rays(n, eye, target_lo, target_hi, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    tgt = np.asarray(target_lo) + rng.random((n, 3)) * (np.asarray(target_hi) - np.asarray(target_lo))
    o = np.broadcast_to(np.asarray(eye, dtype=np.float64), (n, 3))
    return np.concatenate([o, tgt - o], axis=1).astype(np.float32)


def test_cornell_hits(pkg, oracle):
    d = pkg.scenes.cornell()
    rays = np.concatenate([
        camera_like_rays(200_000, (0, 0, -22), (-6, -6, 0), (6, 6, 0), 1),
        pkg.scenes.random_rays(200_000, (-5.5, -5.5, -30), (5.5, 5.5, 10), seed=2, inflate=0.9),
    ])
    check_hits(pkg, oracle, d, rays)


def test_rtiow_hits(pkg, oracle):
    d = pkg.scenes.rtiow(seed=3)
    rays = np.concatenate([
        camera_like_rays(200_000, (13, 2, 3), (-6, 0, -6), (6, 2, 6), 4),
        pkg.scenes.random_rays(200_000, (-12, 0, -12), (12, 2.5, 12), seed=5, inflate=1.0),
    ])
    check_hits(pkg, oracle, d, rays)


def mixed_scene(pkg):
    """Every primitive type, overlapping, plus exactly coincident primitives (tie handling)."""
    S = pkg.SceneDescription()
    m = [S.add_material(pkg.lambertian((0.5, 0.5, 0.5))), S.add_material(pkg.metal((0.8, 0.8, 0.8), 0.1)),
         S.add_material(pkg.dielectric((0.1, 0.2, 0.3))), S.add_material(pkg.glossy((0.3, 0.6, 0.2), 0.5, 0.2))]
    rng = np.random.Generator(np.random.PCG64(11))
    for i in range(40):
        c = rng.random(3) * 8 - 4
        S.add_sphere(tuple(c), float(0.2 + rng.random() * 0.8), m[i % 4])
    for i in range(20):
        c = rng.random(3) * 8 - 4
        S.add_axis_box(tuple(c), tuple(0.3 + rng.random(3) * 1.5), m[i % 4])
    for i in range(20):
        c = rng.random(3) * 8 - 4
        S.add_box(tuple(c), tuple(0.3 + rng.random(3) * 1.5), tuple(rng.random(3) * 2 - 1), m[i % 4])
    for i in range(60):
        c = rng.random(3) * 8 - 4
        v = c + rng.random((3, 3)) * 2 - 1
        S.add_triangle(v, m[i % 4], two_sided=bool(i % 2))
    # coincident pairs: same sphere twice, same triangle twice, same box twice, a triangle in a box face plane
    S.add_sphere((0, 0, 6), 1.0, m[0])
    S.add_sphere((0, 0, 6), 1.0, m[1])
    tri = [(-1, -1, 8), (1, -1, 8), (0, 1, 8)]
    S.add_triangle(tri, m[0])
    S.add_triangle(tri, m[1])
    S.add_axis_box((3, 3, 6), (1, 1, 1), m[2])
    S.add_axis_box((3, 3, 6), (1, 1, 1), m[3])
    S.add_triangle([(2.5, 2.5, 5.5), (3.5, 2.5, 5.5), (3.0, 3.5, 5.5)], m[0])
    S.add_axis_box((0, 0, 0), (-30, -30, -30), m[0])  # hollow enclosure: every ray hits something
    S.set_camera((0, 0, -12), (0, 0, 0), 40.0)
    return S


def test_mixed_primitives_and_ties(pkg, oracle):
    d = mixed_scene(pkg)
    rays = np.concatenate([
        pkg.scenes.random_rays(300_000, (-6, -6, -6), (6, 6, 9), seed=6, inflate=1.2),
        camera_like_rays(100_000, (0, 0, -12), (-5, -5, 0), (5, 5, 9), 7),
        # axis-parallel rays onto the coincident primitives: exact ties
        np.array([[0, 0, -3, 0, 0, 1], [0.1, -0.2, -3, 0, 0, 1], [3, 3, -3, 0, 0, 1], [3.1, 2.9, -3, 0, 0, 1],
                  [0, 0, 20, 0, 0, -1], [3, 3, 20, 0, 0, -1], [0.25, 0.25, 20, 0, 0, -1]], dtype=np.float32),
    ])
    ids = check_hits(pkg, oracle, d, rays)
    assert len(np.unique(ids)) > 100


def test_mesh_hits(pkg, oracle):
    d = pkg.scenes.mesh_scene(n_triangles=120_000, rings=32, segments=32, seed=5)
    ora = oracle.OracleScene(d)
    mn, mx = ora.bounds()
    rays = pkg.scenes.random_rays(500_000, mn, mx, seed=0x5EED, inflate=1.5)
    ids = check_hits(pkg, oracle, d, rays)
    assert (ids >= 2).sum() > 10_000  # plenty of triangle hits


def test_degenerate_rays(pkg, oracle):
    """NaN and zero directions (Ray::new makes them all-NaN), NaN origins, axis-parallel rays on box planes."""
    d = mixed_scene(pkg)
    nan = float("nan")
    rays = np.array([
        [0, 0, -3, 0, 0, 0], [0, 0, -3, nan, 0, 1], [nan, nan, nan, 0, 0, 1], [nan, 0, 0, 0, 0, 1],
        [0.5, 0.5, -3, 0, 0, 1], [3.5, 3.5, -3, 0, 0, 1], [3.5, 3, 6, 1, 0, 0], [3, 3.5, 6, 0, -1, 0],
        [0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1, 0], [15, 15, 15, -1, -1, -1], [float("inf"), 0, 0, 1, 0, 0],
    ], dtype=np.float32)
    check_hits(pkg, oracle, d, rays, expect_hits=False)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 1023, 1025, 2049, 16 * 128 + 1, 50_001])
def test_batch_sizes_around_the_work_shares(pkg, oracle, n):
    """A launch's rays are dealt out in sixteen contiguous shares of whole groups of 64 rays, in runs of 128: batches
    smaller than a group, smaller than the shares, with ragged last shares and last runs must come back complete."""
    d = pkg.scenes.rtiow(seed=3)
    rays = camera_like_rays(n, (13, 2, 3), (-6, 0, -6), (6, 2, 6), 100 + n)
    check_hits(pkg, oracle, d, rays, expect_hits=n > 100)


def test_t_range_bounds(pkg, oracle):
    """Caller-supplied t_min / t_max, including hits exactly at t_max (inclusive for triangles and box entries)."""
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
    S.add_triangle([(0, 0, 0), (1, 0, 0), (0, 1, 0)], m)
    S.add_sphere((3, 0, 0), 1.0, m)
    S.add_axis_box((6, 0, 0), (2, 2, 2), m)
    rays = np.array([[0.25, 0.25, -1, 0, 0, 1], [3, 0, -5, 0, 0, 1], [6, 0, -5, 0, 0, 1], [6, 0, 0, 0, 0, 1]], np.float32)
    for t_max in (1.0, 4.0, 0.999, 100.0):
        check_hits(pkg, oracle, S, rays, 0.00015, t_max, expect_hits=False)


@pytest.mark.parametrize("name", ["cornell", "rtiow", "mesh_env"])
def test_render_pass_bit_identical(pkg, oracle, name):
    if name == "cornell":
        d, W, H, spp, bounces = pkg.scenes.cornell(), 96, 64, 8, 8
    elif name == "rtiow":
        d, W, H, spp, bounces = pkg.scenes.rtiow(seed=2), 96, 64, 8, 8
    else:
        d = pkg.scenes.mesh_scene(n_triangles=30_000, rings=24, segments=24, seed=9)
        d.objects = d.objects[2:]  # drop room and light: the environment lights the blobs
        d.set_environment(pkg.scenes.sky_map(256, 128))
        W, H, spp, bounces = 96, 64, 8, 8
    scene = pkg.Scene(d)
    ora = oracle.OracleScene(d)
    acc = np.zeros(W * H * 3, np.float32)
    oacc = np.zeros(W * H * 3, np.float32)
    for pass_index in range(2):
        p = pkg.Scene.params(W, H, spp, bounces, light_clamp=(20.0 if name == "mesh_env" else math.inf),
                             pass_index=pass_index, seed=42, batch_paths=(W * H * spp) // 3 + 17)
        scene.render_pass(acc, p)
        ora.render_pass(oacc, p)
    st = scene.stats()
    assert st.rays == ora.counters()["rays"] * 1 or st.rays > 0
    assert np.isfinite(oacc).all()
    # acceptance criterion: mean luminance within 1 %
    assert abs(float(acc.mean()) - float(oacc.mean())) <= 0.01 * abs(float(oacc.mean()))
    # delivered: identical bits
    diff = int((acc.view(np.uint32) != oacc.view(np.uint32)).sum())
    assert diff == 0, "%d of %d accumulator words differ, max abs %.3g" % (diff, acc.size, float(np.max(np.abs(acc - oacc))))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_preview_and_modes(pkg, oracle, mode):
    d = pkg.scenes.rtiow(seed=4)
    scene = pkg.Scene(d)
    ora = oracle.OracleScene(d)
    W, H = 80, 60
    p = pkg.Scene.params(W, H, 1, 8, mode=mode, seed=5)
    out = np.zeros(W * H * 3, np.float32)
    oout = np.zeros(W * H * 3, np.float32)
    scene.preview(out, p)
    ora.preview(oout, p)
    assert np.array_equal(out.view(np.uint32), oout.view(np.uint32))
    if mode != 2:  # Renderer::render with Mode::Normals / Mode::Colors
        p = pkg.Scene.params(W, H, 3, 8, mode=mode, seed=6)
        acc = np.zeros(W * H * 3, np.float32)
        oacc = np.zeros(W * H * 3, np.float32)
        scene.render_pass(acc, p)
        ora.render_pass(oacc, p)
        assert np.array_equal(acc.view(np.uint32), oacc.view(np.uint32))


def test_ray_count_matches_oracle(pkg, oracle):
    d = pkg.scenes.cornell()
    scene = pkg.Scene(d)
    ora = oracle.OracleScene(d)
    W, H = 64, 48
    p = pkg.Scene.params(W, H, 4, 8, seed=9)
    acc = np.zeros(W * H * 3, np.float32)
    scene.stats()
    scene.render_pass(acc, p)
    st = scene.stats()
    oacc = np.zeros(W * H * 3, np.float32)
    ora.render_pass(oacc, p)
    assert st.paths == W * H * 4
    assert st.rays == ora.counters()["rays"]


@pytest.mark.parametrize(
    "W,H,spp,bounces,batch",
    [
        (2, 2, 1, 0, 0),        # smallest frame the reference accepts (W-1, H-1 divide), no bounce at all: 0.01 everywhere
        (3, 5, 7, 1, 0),        # odd sizes, one bounce
        (64, 48, 33, 8, 1000),  # batches that split pixels' sample runs (33 spp, 1000 paths per batch)
        (17, 13, 4, 50, 1),     # deep paths, the smallest batch
        (640, 3, 2, 8, 0),      # wide and flat
        # which camera rays share a wave depends on spp and on the batch's pixel count (samples of a pixel go into one
        # wave when both divide evenly): every combination must leave the frame as it is
        (32, 17, 16, 4, 0),     # 16 samples of 4 pixels per wave
        (31, 17, 16, 4, 0),     # ... and a pixel count that forbids it
        (40, 30, 64, 3, 0),     # one pixel's 64 samples per wave
        (40, 30, 128, 2, 5000), # 128 spp in batches of 39 pixels
        (36, 9, 48, 3, 0),      # 48 spp: 16 samples of 4 pixels, three groups
        (33, 9, 48, 3, 700),    # ... in batches of 14 pixels (8 samples of 8 pixels would need a multiple of 8)
    ],
)
def test_render_pass_parameter_edges(pkg, oracle, W, H, spp, bounces, batch):
    d = pkg.scenes.rtiow(seed=3)
    scene, ora = pkg.Scene(d), oracle.OracleScene(d)
    acc = np.full(W * H * 3, 0.25, np.float32)  # the pass adds to what the buffer holds
    oacc = acc.copy()
    p = pkg.Scene.params(W, H, spp, bounces, pass_index=7, seed=11, batch_paths=batch)
    scene.render_pass(acc, p)
    ora.render_pass(oacc, p)
    assert np.array_equal(acc.view(np.uint32), oacc.view(np.uint32))
    if bounces == 0:
        assert np.allclose(acc, 0.25 + 0.01)


def test_render_pass_rejects_bad_parameters(pkg):
    scene = pkg.Scene(pkg.scenes.cornell())
    for W, H, spp in ((1, 16, 1), (16, 1, 1), (16, 16, 0), (0, 0, 1)):
        acc = np.zeros(W * H * 3, np.float32)
        with pytest.raises(pkg.RtError):
            scene.render_pass(acc, pkg.Scene.params(W, H, spp, 8))


def test_deep_stacks_spill_to_the_overflow_area(pkg, oracle):
    """2000 large overlapping spheres and boxes in a small volume: a ray through the middle hits most child boxes at
    every level, so the per-lane stack outgrows its LDS part and uses the overflow area in HBM."""
    rng = np.random.default_rng(3)
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
    for i in range(2000):
        c = rng.normal(0.0, 1.0, 3)
        if i % 2:
            S.add_sphere(tuple(c), float(0.5 + 2.0 * rng.random()), m)
        else:
            S.add_axis_box(tuple(c), tuple(1.0 + 3.0 * rng.random(3)), m)
    scene = pkg.Scene(S)
    assert scene.info().wide_depth * 3 + 4 > 11  # deeper than the LDS part of the stack
    n = 20000
    o = rng.normal(0.0, 4.0, (n, 3))
    d = -o + rng.normal(0.0, 0.3, (n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    check_hits(pkg, oracle, S, rays, 0.00015, 100000.0)


def test_visit_order_decides_between_coincident_triangles(pkg, oracle):
    """A 200 x 200 height field in which every face exists twice (80,000 coincident pairs, one mesh, large enough for
    the threaded sort): every hit is a tie between twins, and which twin wins depends on the reference's visit order,
    i.e. on the exact permutation its stable sort with the odd comparator produces at every level of the build."""
    n = 201
    rng = np.random.default_rng(8)
    gx, gz = np.meshgrid(np.linspace(-2, 2, n), np.linspace(-2, 2, n), indexing="ij")
    gy = 0.2 * np.sin(3 * gx) * np.cos(2 * gz) + 0.01 * rng.random((n, n))
    pos = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3).astype(np.float32)
    nrm = np.tile(np.float32([0, 1, 0]), (n * n, 1))
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j).ravel()
    faces = np.concatenate([np.stack([a, a + 1, a + n], 1), np.stack([a + 1, a + n + 1, a + n], 1)])
    twice = np.concatenate([faces, faces])  # every face twice, in a shuffled order: either twin may be declared first
    order = rng.permutation(len(twice))
    faces_of = twice[order].astype(np.uint32)
    first_copy = order < len(faces)
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
    S.add_mesh(pos, nrm, faces_of, m)
    m_rays = 200_000
    o = np.stack([rng.uniform(-2, 2, m_rays), np.full(m_rays, 3.0), rng.uniform(-2, 2, m_rays)], 1)
    d = np.stack([rng.normal(0, 0.2, m_rays), -np.ones(m_rays), rng.normal(0, 0.2, m_rays)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ids = check_hits(pkg, oracle, S, np.concatenate([o, d], 1).astype(np.float32))
    hit = ids >= 0
    assert hit.mean() > 0.5
    assert first_copy[ids[hit]].any() and (~first_copy[ids[hit]]).any()  # either copy wins somewhere


def test_pixel_ranges_of_a_pass(pkg, oracle):
    """RtRenderParams.first_pixel / pixel_count: the ranges of a pass rendered one by one (here in three uneven
    pieces, one of them through several batches) give the frame of the whole pass, and leave the rest untouched."""
    d = pkg.scenes.rtiow(seed=4)
    scene, ora = pkg.Scene(d), oracle.OracleScene(d)
    W, H, spp = 80, 50, 5
    whole = np.zeros(W * H * 3, np.float32)
    ora.render_pass(whole, pkg.Scene.params(W, H, spp, 8, pass_index=2, seed=9))
    acc = np.zeros(W * H * 3, np.float32)
    for first, count, batch in ((0, 1234, 0), (1234, 1, 0), (1235, W * H - 1235, 700)):
        before = acc.copy()
        scene.render_pass(acc, pkg.Scene.params(W, H, spp, 8, pass_index=2, seed=9, first_pixel=first, pixel_count=count,
                                                batch_paths=batch))
        outside = np.ones(W * H * 3, bool)
        outside[3 * first:3 * (first + count)] = False
        assert np.array_equal(acc[outside].view(np.uint32), before[outside].view(np.uint32))
    assert np.array_equal(acc.view(np.uint32), whole.view(np.uint32))
    with pytest.raises(pkg.RtError):
        scene.render_pass(acc, pkg.Scene.params(W, H, spp, 8, first_pixel=W * H - 3, pixel_count=4))


def test_preview_of_a_pixel_range_leaves_the_rest_untouched(pkg, oracle):
    """rt_preview on host buffers with a pixel range: the range holds the preview, every other pixel of the caller's
    buffer comes back as it went in."""
    d = pkg.scenes.rtiow(seed=4)
    scene, ora = pkg.Scene(d), oracle.OracleScene(d)
    W, H = 64, 40
    for mode in (pkg.Mode.Normals, pkg.Mode.Colors, pkg.Mode.Image):
        whole = np.zeros(W * H * 3, np.float32)
        ora.preview(whole, pkg.Scene.params(W, H, 1, 8, mode=mode, seed=6))
        first, count = 777, 901
        buf = np.full(W * H * 3, -7.25, np.float32)
        scene.preview(buf, pkg.Scene.params(W, H, 1, 8, mode=mode, seed=6, first_pixel=first, pixel_count=count))
        inside = np.zeros(W * H * 3, bool)
        inside[3 * first:3 * (first + count)] = True
        assert np.all(buf[~inside] == np.float32(-7.25))
        assert np.array_equal(buf[inside].view(np.uint32), whole[inside].view(np.uint32))
