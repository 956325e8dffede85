"""Wavefront OBJ reader with the behaviour the reference asks of tobj (src/io.rs:164-172):
one model per `o`/`g` statement, a unified index over distinct (v, vt, vn) corners, fan
triangulation, points and lines dropped.  Python twin of host/obj.hpp."""
import numpy as np


def load(path):
    v, vn, n_vt = [], [], 0
    models = []
    cur = dict(name="unnamed_object", pos=[], nrm=[], idx=[], has_normals=True)
    remap = {}

    def flush():
        nonlocal cur, remap
        if cur["idx"]:
            models.append(dict(
                name=cur["name"],
                positions=np.array(cur["pos"], np.float32).reshape(-1, 3),
                normals=np.array(cur["nrm"], np.float32).reshape(-1, 3) if cur["has_normals"] else None,
                indices=np.array(cur["idx"], np.uint32).reshape(-1, 3)))
        cur = dict(name=cur["name"], pos=[], nrm=[], idx=[], has_normals=True)
        remap = {}

    with open(path) as f:
        for line in f:
            parts = line.split()
            if not parts or parts[0].startswith("#"):
                continue
            tag = parts[0]
            if tag == "v":
                v.append([float(x) for x in parts[1:4]])
            elif tag == "vn":
                vn.append([float(x) for x in parts[1:4]])
            elif tag == "vt":
                n_vt += 1
            elif tag in ("o", "g"):
                flush()
                cur["name"] = line.split(None, 1)[1].strip() if len(parts) > 1 else "unnamed_object"
            elif tag == "usemtl":
                if cur["idx"]:
                    flush()
            elif tag == "f":
                corner = []
                for tok in parts[1:]:
                    fields = tok.split("/")
                    iv = int(fields[0])
                    it = int(fields[1]) if len(fields) > 1 and fields[1] else 0
                    inn = int(fields[2]) if len(fields) > 2 and fields[2] else 0
                    if iv < 0:
                        iv = len(v) + iv + 1
                    if inn < 0:
                        inn = len(vn) + inn + 1
                    if it < 0:
                        it = n_vt + it + 1
                    key = (iv, it, inn)
                    if key not in remap:
                        remap[key] = len(cur["pos"])
                        cur["pos"].append(v[iv - 1])
                        if 0 < inn <= len(vn):
                            cur["nrm"].append(vn[inn - 1])
                        else:
                            cur["has_normals"] = False
                            cur["nrm"].append([0.0, 0.0, 0.0])
                    corner.append(remap[key])
                for i in range(1, len(corner) - 1):
                    cur["idx"].append([corner[0], corner[i], corner[i + 1]])
    flush()
    return models
