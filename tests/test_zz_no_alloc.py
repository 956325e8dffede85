"""The closest-hit entry point that allocates nothing (rt_closest_hit_batch_noalloc_device) against the ordinary
device entry point: identical ids, distances, normals and points, on two calls in a row, the second smaller (the
shared per-GPU work array keeps the first call's contents), and a clear error beyond the size limit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(pkg, scene, rays):
    import torch

    n = len(rays)
    d_rays = torch.from_numpy(rays).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    out = []
    for fn in (scene.closest_hit_device, scene.closest_hit_noalloc_device):
        ids = torch.full((n,), -5, dtype=torch.int32, device="cuda")
        t = torch.full((n,), -5.0, dtype=torch.float32, device="cuda")
        nrm = torch.full((n, 3), -5.0, dtype=torch.float32, device="cuda")
        pt = torch.full((n, 3), -5.0, dtype=torch.float32, device="cuda")
        fn(d_rays.data_ptr(), n, ids.data_ptr(), t.data_ptr(), nrm.data_ptr(), pt.data_ptr(), stream)
        torch.cuda.synchronize()
        out.append((ids.cpu().numpy(), t.cpu().numpy(), nrm.cpu().numpy(), pt.cpu().numpy()))
    return out


def test_noalloc_closest_hit_matches_the_allocating_entry_point(pkg):
    lib = pkg.load()
    pkg._abi.check(lib.rt_init(0))
    d = pkg.scenes.mesh_scene(n_triangles=120_000, rings=32, segments=32, seed=5)
    scene = pkg.Scene(d)
    info = scene.info()
    hits = 0
    for n, seed in ((pkg._abi.RT_NOALLOC_MAX_RAYS, 31), (70_001, 32)):  # the second call is smaller
        rays = pkg.scenes.random_rays(n, info.bounds_min[:], info.bounds_max[:], seed=seed)
        rays[5] = (0, 0, 0, 0, 0, 0)  # a zero direction: Ray::new makes it NaN, the answer is a miss
        (ids, t, nrm, pt), (ids2, t2, nrm2, pt2) = _both(pkg, scene, rays)
        assert np.array_equal(ids, ids2)
        assert np.array_equal(t.view(np.uint32), t2.view(np.uint32))
        assert np.array_equal(nrm.view(np.uint32), nrm2.view(np.uint32))
        assert np.array_equal(pt.view(np.uint32), pt2.view(np.uint32))
        assert ids2[5] == -1 and ids2.min() >= -1
        hits += int((ids2 >= 0).sum())
    assert hits > 50_000
    # outputs may be left out, zero rays is a no-op, more than the limit is refused
    import torch

    rays = torch.from_numpy(pkg.scenes.random_rays(1000, info.bounds_min[:], info.bounds_max[:], seed=3)).cuda()
    ids = torch.empty(1000, dtype=torch.int32, device="cuda")
    scene.closest_hit_noalloc_device(rays.data_ptr(), 1000, ids.data_ptr(), None, None, None, 0)
    scene.closest_hit_noalloc_device(rays.data_ptr(), 0, None, None, None, None, 0)
    torch.cuda.synchronize()
    ref = torch.empty(1000, dtype=torch.int32, device="cuda")
    scene.closest_hit_device(rays.data_ptr(), 1000, ref.data_ptr(), None, None, None, 0)
    torch.cuda.synchronize()
    assert torch.equal(ids, ref)
    with pytest.raises(pkg.RtError) as e:
        scene.closest_hit_noalloc_device(rays.data_ptr(), pkg._abi.RT_NOALLOC_MAX_RAYS + 1, ids.data_ptr(), None, None, None, 0)
    assert "at most" in str(e.value)


def _chain_scene(pkg, n=200):
    """Spheres whose centres grow geometrically along x: the reference's middle split (src/tracer/bvh.rs:106-117) peels
    them off one or two at a time, so the tree is about as deep as it is large."""
    s = pkg.scene.SceneDescription()
    s.set_camera(eye=(-5.0, 0.0, 0.0), lookat=(0.0, 0.0, 0.0), fov=40.0, focus_dist=5.0, aperture=0.0)
    m = s.add_material(pkg.scene.lambertian((0.6, 0.6, 0.6)))
    for k in range(n):
        x = 1.5 ** k
        s.add_sphere((x, 0.0, 0.0), 0.2 * x, m)
    return s


def test_noalloc_deep_tree_shrinks_the_grid_and_allocates_nothing(pkg, monkeypatch):
    """A tree deep enough that the traversal stacks' overflow area for a full grid would not fit in the work array (the
    entry point then launches fewer blocks), through both node forms; the device's free memory does not move across
    the calls (no hipMalloc / hipFree behind them)."""
    import torch

    lib = pkg.load()
    pkg._abi.check(lib.rt_init(0))
    d = _chain_scene(pkg)
    scenes = [pkg.Scene(d)]
    monkeypatch.setenv("RTCORE_NODES", "wide")
    scenes.append(pkg.Scene(d))
    monkeypatch.delenv("RTCORE_NODES")
    assert [s.info().node_bytes for s in scenes] == [64, 128]
    # 3 stack entries per level + 4, 11 of them in LDS, 2 KiB per entry and block: beyond 27 entries a grid of
    # 7 blocks on each of 256 CUs no longer fits in the 53 MiB the work array leaves for it
    assert scenes[0].info().wide_depth >= 40
    n = 100_000
    rng = np.random.default_rng(5)
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0] = -5.0
    rays[:, 1:3] = rng.uniform(-0.3, 0.3, (n, 2))
    rays[:, 3] = 1.0
    rays[:, 4:6] = rng.uniform(-0.05, 0.05, (n, 2))
    d_rays = torch.from_numpy(rays).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    out = [[torch.empty(n, dtype=torch.int32, device="cuda"), torch.empty(n, dtype=torch.float32, device="cuda")] for _ in range(3)]
    scenes[0].closest_hit_device(d_rays.data_ptr(), n, out[0][0].data_ptr(), out[0][1].data_ptr(), None, None, stream)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for k, scene in enumerate(scenes):
        scene.closest_hit_noalloc_device(d_rays.data_ptr(), n, out[1 + k][0].data_ptr(), out[1 + k][1].data_ptr(), None, None, stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    for k in (1, 2):
        assert torch.equal(out[0][0], out[k][0])
        assert torch.equal(out[0][1].view(torch.int32), out[k][1].view(torch.int32))
    assert int((out[0][0] >= 0).sum()) > n // 2
