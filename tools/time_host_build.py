import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_package(); lib = pkg.load()
n = int(sys.argv[1])
d = pkg.scenes.mesh_scene(n_triangles=n, seed=7)
desc = d.to_desc(); info = pkg._abi.RtSceneInfo()
t0 = time.time(); rc = lib.rt_scene_validate(C.byref(desc), C.byref(info)); print("rc", rc, "validate %.2f s" % (time.time() - t0), info.n_wide_nodes, info.wide_depth)
