import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
d = pkg.scenes.mesh_scene(n_triangles=10_000_000, seed=7)
desc = d.to_desc()
info = pkg._abi.RtSceneInfo()
print(pkg.load().rt_scene_validate(C.byref(desc), C.byref(info)), info.n_wide_nodes)
