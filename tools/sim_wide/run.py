#!/usr/bin/env python3
"""Builds tools/sim_wide/sim.cpp against the library's host build and prints what an 8-wide form of the tree would cost
per ray next to the 4-wide form (CPU only; see sim.cpp).  Usage: python tools/sim_wide/run.py [--tris N] [--rays N] [--width 8]"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--rays", type=int, default=400_000)
    ap.add_argument("--width", type=int, default=8)
    args = ap.parse_args()
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rtcore_sim_wide")
    os.makedirs(out, exist_ok=True)
    lib_path = os.path.join(out, "libsim.so")
    csrc = os.path.join(ROOT, "rust-raytracer_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-mavx2", "-mfma", "-I" + csrc,
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "sim_wide", "sim.cpp"),
                           os.path.join(csrc, "scene_build.cpp"), "-o", lib_path, "-lpthread"])
    import __graft_entry__ as ge

    pkg = ge.load_package()
    desc = pkg.scenes.mesh_scene(n_triangles=args.tris, seed=7)
    d = desc.to_desc()
    lib = C.CDLL(lib_path)
    res = (C.c_double * 96)()
    nodes = (C.c_double * 3)()
    lib.sim_run.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rc = lib.sim_run(C.cast(C.byref(d), C.c_void_p), args.rays, 1, args.width, res, nodes)
    if rc:
        raise SystemExit(rc)
    print("%d triangles, %d rays; 4-wide nodes %d, %d-wide nodes %d (%.2f children per node)" % (
        desc.num_primitives(), args.rays, nodes[0], args.width, nodes[1], nodes[2]))
    names = ["4-wide, sorted", "%d-wide, sorted" % args.width, "%d-wide, octant order" % args.width, "4-wide, octant order"]
    print("%-24s %8s %8s %8s %8s %8s %6s %8s %6s" % ("form", "nodes", "leaves", "tests", "pushes", "depth", "max", "boxes", "hit"))
    for m, name in enumerate(names):
        r = res[m * 8:(m + 1) * 8]
        print("%-24s %8.2f %8.2f %8.2f %8.2f %8.2f %6d %8.1f %6.2f" % (name, r[0], r[1], r[2], r[3], r[4], int(r[5]), r[6], r[7]))
    print("share of node steps with more than k stack entries (k = 0 .. 15):")
    for m, name in enumerate(names):
        print("%-24s %s" % (name, " ".join("%.3f" % res[32 + m * 16 + k] for k in range(16))))


if __name__ == "__main__":
    main()
