"""The opt-in scene build on the GPU (RT_BUILD_DEVICE) against the CPU oracle.

The device build cannot reproduce the reference's visit order (src/tracer/bvh.rs:39-117 sorts sequentially with a
comparator that is not a strict weak order), so what is pinned here is what it does promise: the same primitive records,
hence the same (prim_id, t, normal, point) for every ray whose closest hit is not an exact tie between two primitives.
Mismatches are counted, and on the scene made of coincident twins every mismatch must be such a tie."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _compare(pkg, oracle, desc, rays):
    """Returns (mismatching ray indices, device-build results, oracle results)."""
    scene = pkg.Scene(desc, builder=pkg._abi.RT_BUILD_DEVICE)
    info = scene.info()
    assert info.builder == pkg._abi.RT_BUILD_DEVICE and info.n_primitives == desc.num_primitives()
    ora = oracle.OracleScene(desc)
    got = scene.closest_hit(rays)
    want = ora.closest_hit(rays)
    scene.close()
    bad = np.zeros(len(rays), bool)
    for g, w in zip(got, want):
        diff = g.view(np.uint32).reshape(len(rays), -1) != w.view(np.uint32).reshape(len(rays), -1)
        bad |= diff.any(axis=1)
    return np.nonzero(bad)[0], got, want


def test_mesh_scene_hits_match_the_oracle(pkg, oracle):
    d = pkg.scenes.mesh_scene(n_triangles=300_000, rings=32, segments=32, seed=5)
    mn, mx = oracle.OracleScene(d).bounds()
    rays = np.concatenate([pkg.scenes.random_rays(700_000, mn, mx, seed=0x5EED, inflate=1.5),
                           pkg.scenes.random_rays(300_000, mn, mx, seed=3, inflate=0.7)])
    bad, got, want = _compare(pkg, oracle, d, rays)
    assert (want[0] >= 2).sum() > 100_000  # plenty of triangle hits
    # t, normal-independent fields: identical bits for every ray.  A ray that meets the shared edge of two triangles
    # at exactly the same t on both is a tie (about one ray in a few million here): which of the two is named follows
    # this tree's order, and the normal with it.  Nothing else may differ.
    assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)), "t differs"
    assert np.array_equal(got[3].view(np.uint32), want[3].view(np.uint32)), "hit point differs"
    ties = bad[got[0][bad] != want[0][bad]]
    assert len(ties) == len(bad) and len(bad) <= 5, "%d of %d rays differ (ids differ on %d)" % (len(bad), len(rays), len(ties))
    print("mesh scene: %d exact-tie rays of %d name the neighbouring triangle" % (len(bad), len(rays)))


def test_every_primitive_type_and_tiny_scenes(pkg, oracle):
    from test_gpu_parity import camera_like_rays, mixed_scene

    d = mixed_scene(pkg)
    rays = np.concatenate([pkg.scenes.random_rays(300_000, (-6, -6, -6), (6, 6, 9), seed=6, inflate=1.2),
                           camera_like_rays(100_000, (0, 0, -12), (-5, -5, 0), (5, 5, 9), 7)])
    bad, got, want = _compare(pkg, oracle, d, rays)
    # the scene holds exactly coincident pairs: a mismatch must be a tie between two primitives at the same t
    assert np.array_equal(got[1][bad].view(np.uint32), want[1][bad].view(np.uint32))
    assert len(bad) < 0.02 * len(rays)
    for n_prims in (1, 2, 3, 5):
        S = pkg.SceneDescription()
        m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
        for k in range(n_prims):
            S.add_sphere((2.5 * k - 3.0, 0.1 * k, 0.0), 1.0, m)
        r = pkg.scenes.random_rays(20_000, (-5, -2, -2), (9, 2, 2), seed=n_prims, inflate=1.3)
        bad, got, want = _compare(pkg, oracle, S, r)
        assert len(bad) == 0 and (want[0] >= 0).sum() > 100


def test_cornell_rtiow_teaset_passes_are_bit_identical(pkg, oracle):
    W, H = 96, 64
    for d in (pkg.scenes.cornell(), pkg.scenes.rtiow(seed=3), pkg.scenes.teaset(env_size=(512, 256))):
        scene = pkg.Scene(d, builder=pkg._abi.RT_BUILD_DEVICE)
        ora = oracle.OracleScene(d)
        p = pkg.Scene.params(W, H, 8, 8, seed=21)
        acc, oacc = np.zeros(W * H * 3, np.float32), np.zeros(W * H * 3, np.float32)
        scene.render_pass(acc, p)
        ora.render_pass(oacc, p)
        differing = int((acc.view(np.uint32) != oacc.view(np.uint32)).reshape(-1, 3).any(axis=1).sum())
        assert differing == 0, "%d of %d pixels differ" % (differing, W * H)
        scene.close()


def test_coincident_twins_differ_only_in_which_twin_is_named(pkg, oracle):
    """Every face exists twice: every hit is an exact tie, and which twin wins follows the tree's own order."""
    n = 101
    rng = np.random.default_rng(8)
    gx, gz = np.meshgrid(np.linspace(-2, 2, n), np.linspace(-2, 2, n), indexing="ij")
    gy = 0.2 * np.sin(3 * gx) * np.cos(2 * gz) + 0.01 * rng.random((n, n))
    pos = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3).astype(np.float32)
    nrm = np.tile(np.float32([0, 1, 0]), (n * n, 1))
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j).ravel()
    faces = np.concatenate([np.stack([a, a + 1, a + n], 1), np.stack([a + 1, a + n + 1, a + n], 1)])
    order = rng.permutation(2 * len(faces))
    faces_of = np.concatenate([faces, faces])[order].astype(np.uint32)
    S = pkg.SceneDescription()
    S.add_mesh(pos, nrm, faces_of, S.add_material(pkg.lambertian((0.5, 0.5, 0.5))))
    m_rays = 100_000
    o = np.stack([rng.uniform(-2, 2, m_rays), np.full(m_rays, 3.0), rng.uniform(-2, 2, m_rays)], 1)
    dd = np.stack([rng.normal(0, 0.2, m_rays), -np.ones(m_rays), rng.normal(0, 0.2, m_rays)], 1)
    rays = np.concatenate([o, dd / np.linalg.norm(dd, axis=1, keepdims=True)], 1).astype(np.float32)
    bad, got, want = _compare(pkg, oracle, S, rays)
    hit = want[0] >= 0
    assert hit.mean() > 0.5
    # same t, same normal, same point everywhere; the id differs only between twins (same three vertices)
    for k in (1, 2, 3):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32))
    assert np.array_equal(got[0] >= 0, hit)
    g, w = got[0][bad], want[0][bad]
    assert np.array_equal(np.sort(faces_of[g], axis=1), np.sort(faces_of[w], axis=1))
    print("coincident twins: %d of %d hits name the other twin" % (len(bad), int(hit.sum())))


def test_morton_order_tree_gives_the_same_hits(pkg, oracle, monkeypatch):
    """RTCORE_DEVICE_TREE=morton: the other binary tree of the device build (Morton codes, radix sort, binary radix
    tree).  Same records and guard boxes, so the same hits; only the number of box tests differs."""
    if b"+experiments" not in pkg.load().rt_version():
        pytest.skip("the Morton-order tree is compiled only into `make experiments` builds (load one through RTCORE_LIB)")
    monkeypatch.setenv("RTCORE_DEVICE_TREE", "morton")
    d = pkg.scenes.mesh_scene(n_triangles=100_000, rings=24, segments=24, seed=11)
    mn, mx = oracle.OracleScene(d).bounds()
    rays = pkg.scenes.random_rays(300_000, mn, mx, seed=17, inflate=1.2)
    bad, got, want = _compare(pkg, oracle, d, rays)
    assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)) and len(bad) <= 3


def test_bad_vertex_index_and_degenerate_input(pkg):
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
    S.add_mesh(np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32), np.array([[0, 1, 7]], np.uint32), m)
    with pytest.raises(pkg.RtError) as e:
        pkg.Scene(S, builder=pkg._abi.RT_BUILD_DEVICE)
    assert e.value.code == pkg._abi.RT_ERR_INVALID_ARGUMENT and "vertex index" in str(e.value)
    # 5000 identical triangles and one with a NaN vertex (a smaller box beside 5000 equal ones: the reference would split
    # the equal ones off one by one, 5000 levels deep): the build terminates with a shallow tree and the scene answers
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
    pos = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [np.nan, 0, 0]])
    idx = np.concatenate([np.tile(np.uint32([[0, 1, 2]]), (5000, 1)), np.uint32([[0, 1, 3]])])
    S.add_mesh(pos, np.tile(np.float32([0, 0, 1]), (4, 1)), idx, m)
    scene = pkg.Scene(S, builder=pkg._abi.RT_BUILD_DEVICE)
    assert scene.info().wide_depth < 40
    ids, t, _, _ = scene.closest_hit(np.float32([[0.2, 0.2, -1, 0, 0, 1], [5, 5, -1, 0, 0, 1]]))
    assert 0 <= ids[0] < 5000 and t[0] == 1.0 and ids[1] == -1
    scene.close()


def test_build_time_one_million_triangles(pkg):
    d = pkg.scenes.mesh_scene(n_triangles=1_000_000, seed=7)
    d.to_desc()
    t0 = time.time()
    a = pkg.Scene(d, builder=pkg._abi.RT_BUILD_DEVICE)
    t_dev = time.time() - t0
    t0 = time.time()
    b = pkg.Scene(d, builder=pkg._abi.RT_BUILD_EXACT)
    t_host = time.time() - t0
    ia, ib = a.info(), b.info()
    print("1 M triangles: device build %.0f ms (library clock %.0f ms), %d wide nodes, depth %d; host build %.0f ms, %d wide nodes, depth %d"
          % (t_dev * 1e3, ia.build_ms, ia.n_wide_nodes, ia.wide_depth, t_host * 1e3, ib.n_wide_nodes, ib.wide_depth))
    assert ia.n_primitives == ib.n_primitives and ia.build_ms < ib.build_ms
    a.close()
    b.close()
