#!/usr/bin/env python3
"""Builds rust-raytracer_amd/data/teaset.npz from the reference's OBJ data files
(scene/models/newell_teaset/*.obj, the Newell tea set).  Run in the container that has
/root/reference; the GPU box only sees the committed .npz.  The fixture holds, per tobj
model, the unified positions / normals / indices exactly as objio.load returns them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/scene/models/newell_teaset"
out = {}
for name in ("teapot", "teacup", "cup", "spoon"):
    models = pkg.objio.load(os.path.join(src, name + ".obj"))
    out[name + "_count"] = np.array(len(models))
    for i, m in enumerate(models):
        out["%s_%d_positions" % (name, i)] = m["positions"]
        out["%s_%d_normals" % (name, i)] = m["normals"]
        out["%s_%d_indices" % (name, i)] = m["indices"]
    print(name, len(models), "models", sum(len(m["indices"]) for m in models), "triangles")
path = os.path.join(ROOT, "rust-raytracer_amd", "data", "teaset.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
