#!/usr/bin/env python3
"""Configuration C1: the reference's default procedural spheres scene, 640x480, 128 spp, 8 bounces, on the CPU
oracle only (plumbing, no GPU).  Writes profiles/r01_c1_cpu_oracle.json: s/frame, Mrays/s, mean luminance, checksum."""
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import oracle_binding as ob  # noqa: E402

pkg = ge.load_package()
W, H, spp, bounces = 640, 480, 128, 8
threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
ora = ob.OracleScene(pkg.scenes.rtiow(seed=1))
acc = np.zeros(W * H * 3, np.float32)
t0 = time.perf_counter()
ora.render_pass(acc, pkg.Scene.params(W, H, spp, bounces, seed=1), threads=threads)
dt = time.perf_counter() - t0
c = ora.counters()
lum = 0.2126 * acc[0::3] + 0.7152 * acc[1::3] + 0.0722 * acc[2::3]
res = {"config": "C1 procedural RTIOW scene (seed 1), %dx%d, %d spp, %d bounces, CPU oracle" % (W, H, spp, bounces),
       "threads": threads, "s_per_frame": round(dt, 3), "Mrays_per_s": round(c["rays"] / dt / 1e6, 3), "rays": c["rays"],
       "mean_luminance": float(lum.mean()), "crc32_of_accumulator": zlib.crc32(acc.tobytes())}
json.dump(res, open(os.path.join(ROOT, "profiles", "r01_c1_cpu_oracle.json"), "w"), indent=1)
print(json.dumps(res))
