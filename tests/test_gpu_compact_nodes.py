"""The compact form of the tree (64-byte nodes on a 256-step grid, exact guard box in each leaf block) against the CPU
oracle where that form could go wrong: rays that run along an axis (1/d infinite on the other axes), origins exactly
on box planes, flat nodes, scenes at very small and very large scale, and the 128-byte nodes kept as the other form.
Everything here is bit for bit, like the rest of the GPU parity tests."""
import time

import numpy as np
import pytest

from test_gpu_parity import check_hits

pytestmark = pytest.mark.gpu


def _lattice_scene(pkg, n=65, spheres=6):
    """A height field whose vertices lie on multiples of 1/8 (so box planes do too, and many nodes are flat on y),
    spheres on an integer lattice above it and an axis-aligned box around everything."""
    rng = np.random.default_rng(5)
    gx, gz = np.meshgrid(np.arange(n) / 8.0 - (n - 1) / 16.0, np.arange(n) / 8.0 - (n - 1) / 16.0, indexing="ij")
    gy = np.round(4.0 * np.sin(1.3 * gx) * np.cos(0.9 * gz)) / 8.0  # steps of 1/8: large flat patches
    pos = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3).astype(np.float32)
    nrm = np.tile(np.float32([0, 1, 0]), (n * n, 1))
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j).ravel()
    faces = np.concatenate([np.stack([a, a + 1, a + n], 1), np.stack([a + 1, a + n + 1, a + n], 1)]).astype(np.uint32)
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
    S.add_mesh(pos, nrm, faces[rng.permutation(len(faces))], m, cull_backface=False)
    for x in range(-spheres // 2, spheres // 2 + 1):
        for z in range(-spheres // 2, spheres // 2 + 1):
            S.add_sphere((float(x), 2.0, float(z)), 0.5, m)
    S.add_axis_box((0.0, 1.0, 0.0), (-12.0, -8.0, -12.0), m)  # hollow room
    return S


def _lattice_rays(n, seed):
    """Origins on multiples of 1/8; directions along axes, along face diagonals (one component exactly zero), random
    with one component zeroed, and random."""
    rng = np.random.default_rng(seed)
    o = np.round(rng.uniform(-5.0, 5.0, (n, 3)) * 8.0) / 8.0
    o[:, 1] = np.round(rng.uniform(-2.5, 4.0, n) * 8.0) / 8.0
    d = rng.normal(size=(n, 3))
    kind = rng.integers(0, 4, n)
    axis = rng.integers(0, 3, n)
    sign = rng.choice([-1.0, 1.0], n)
    along = kind == 0
    d[along] = 0.0
    d[along, axis[along]] = sign[along]
    diag = kind == 1
    d[diag] = rng.choice([-1.0, 1.0], (int(diag.sum()), 3))
    d[diag, axis[diag]] = 0.0
    one_zero = kind == 2
    d[one_zero, axis[one_zero]] = 0.0
    return np.concatenate([o, d], axis=1).astype(np.float32)


def test_rays_along_axes_and_origins_on_box_planes(pkg, oracle):
    S = _lattice_scene(pkg)
    rays = _lattice_rays(400_000, 41)
    ids = check_hits(pkg, oracle, S, rays)
    assert (ids >= 0).mean() > 0.9
    # negative zero components and a negative t_min go the same way
    rays2 = rays[:100_000].copy()
    rays2[:, 3:][rays2[:, 3:] == 0.0] = -0.0
    check_hits(pkg, oracle, S, rays2, t_min=-3.0, t_max=50.0)


def test_axis_parallel_rays_do_not_walk_the_whole_tree(pkg):
    """A ray along an axis has an infinite 1/d on the other two: those axes must still turn the ray away from the boxes
    beside its line (ignoring them would send each such ray through most of a 1 M-triangle tree)."""
    d = pkg.scenes.mesh_scene(n_triangles=1_000_000, seed=7)
    scene = pkg.Scene(d)
    rng = np.random.default_rng(3)
    n = 2_000_000
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0] = rng.uniform(-8, 8, n)
    rays[:, 2] = rng.uniform(-8, 8, n)
    rays[:, 1] = 11.0
    rays[:, 4] = -1.0
    scene.closest_hit(rays)  # work buffers
    t0 = time.time()
    ids, t, _, _ = scene.closest_hit(rays)
    elapsed = time.time() - t0
    scene.close()
    print("%d rays straight down a 1 M-triangle scene: %.3f s" % (n, elapsed))
    assert (ids >= 0).all() and elapsed < 3.0


@pytest.mark.parametrize("exponent", [-50, -20, 40, 55])
def test_scenes_at_extreme_scales(pkg, oracle, exponent):
    """Spheres and boxes at 2^exponent times the unit scene (triangles drop out below the reference's 1e-6 determinant
    bound, src/tracer/triangle.rs:46): grid steps far from 1 and origins far from the grid's range."""
    k = float(2.0 ** exponent)
    rng = np.random.default_rng(abs(exponent) + 100)
    S = pkg.SceneDescription()
    m = S.add_material(pkg.lambertian((0.5, 0.5, 0.5)))
    for _ in range(400):
        c = rng.uniform(-10, 10, 3)
        S.add_sphere(tuple(float(np.float32(v * k)) for v in c), float(np.float32(rng.uniform(0.2, 0.8) * k)), m)
    for _ in range(100):
        c = rng.uniform(-10, 10, 3)
        S.add_axis_box(tuple(float(np.float32(v * k)) for v in c), tuple(float(np.float32(v * k)) for v in rng.uniform(0.2, 1.5, 3)), m)
    rays = pkg.scenes.random_rays(200_000, (-12 * k,) * 3, (12 * k,) * 3, seed=abs(exponent) + 7, inflate=1.3)
    rays[::7, 3] = 0.0  # some rays in a coordinate plane
    ids = check_hits(pkg, oracle, S, rays, t_min=0.00015 * k, t_max=1.0e5 * k)
    assert (ids >= 0).mean() > 0.05


def test_wide_nodes_kept_on_request(pkg, oracle, monkeypatch):
    """RTCORE_NODES=wide: the closest-hit kernel reads the 128-byte nodes with exact child boxes instead."""
    monkeypatch.setenv("RTCORE_NODES", "wide")
    d = pkg.scenes.mesh_scene(n_triangles=150_000, rings=32, segments=32, seed=9)
    scene = pkg.Scene(d)
    assert scene.info().node_bytes == 128
    scene.close()
    rays = pkg.scenes.random_rays(300_000, *oracle.OracleScene(d).bounds(), seed=12, inflate=1.2)
    check_hits(pkg, oracle, d, rays)
    monkeypatch.delenv("RTCORE_NODES")
    scene = pkg.Scene(d)
    assert scene.info().node_bytes == 64
    scene.close()
