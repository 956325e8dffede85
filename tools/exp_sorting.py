#!/usr/bin/env python3
"""Experiment: how much does ray order matter to the closest-hit kernel?  Secondary-like rays (origins on
surfaces, uniform directions) traced in random order, in generation order, and sorted by direction octant +
Morton code of the origin."""
import math, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); lib = pkg.load(); pkg._abi.check(lib.rt_init(0))
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scene = pkg.Scene(pkg.scenes.mesh_scene(n_triangles=tris, seed=7)); info = scene.info()
n = 32_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
lo = torch.tensor(info.bounds_min[:], device="cuda") * 0.6; hi = torch.tensor(info.bounds_max[:], device="cuda") * 0.6
def rand_dirs(n):
    z = torch.rand(n, device="cuda", generator=g) * 2 - 1; ph = torch.rand(n, device="cuda", generator=g) * 2 * math.pi
    r = torch.sqrt(torch.clamp(1 - z * z, min=0)); return torch.stack([r * torch.cos(ph), r * torch.sin(ph), z], 1)
org = lo + torch.rand((n, 3), device="cuda", generator=g) * (hi - lo)
rays = torch.cat([org, rand_dirs(n)], 1).float().contiguous()
ids = torch.empty(n, dtype=torch.int32, device="cuda"); t = torch.empty(n, device="cuda"); pt = torch.empty((n, 3), device="cuda")
st = torch.cuda.current_stream().cuda_stream
scene.closest_hit_device(rays.data_ptr(), n, ids.data_ptr(), t.data_ptr(), None, pt.data_ptr(), st); torch.cuda.synchronize()
hit = ids >= 0
o2 = pt[hit]; m = o2.shape[0]
sec = torch.cat([o2, rand_dirs(m)], 1).float().contiguous()
print("secondary rays", m)
def bench(r, label):
    r = r.contiguous(); k = r.shape[0]
    scene.stats(); scene.set_stage_timing(True)
    for _ in range(3):
        scene.closest_hit_device(r.data_ptr(), k, ids.data_ptr(), t.data_ptr(), None, None, st)
    torch.cuda.synchronize(); s = scene.stats(); scene.set_stage_timing(False)
    print("%-28s %8.1f Mrays/s (extend only)" % (label, 3 * k / (s.extend_ms / 1e3) / 1e6))
bench(sec, "generation order")
perm = torch.randperm(m, device="cuda", generator=g)
bench(sec[perm], "random order")
def morton(o):
    q = ((o - o.min(0).values) / (o.max(0).values - o.min(0).values + 1e-9) * 1023).long().clamp(0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
d = sec[:, 3:6]; octant = ((d[:, 0] < 0).long() | ((d[:, 1] < 0).long() << 1) | ((d[:, 2] < 0).long() << 2))
key = (octant << 30) | morton(sec[:, 0:3])
t0 = time.time(); order = torch.argsort(key); torch.cuda.synchronize(); print("torch sort %.1f ms" % ((time.time() - t0) * 1e3))
bench(sec[order], "octant + morton sorted")
order2 = torch.argsort(morton(sec[:, 0:3])); bench(sec[order2], "morton(origin) sorted")
order3 = torch.argsort(octant); bench(sec[order3], "octant only (stable? no)")
