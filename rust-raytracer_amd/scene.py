"""Host-side scene description: the objects and materials of the reference's scene
files (src/io.rs:24-94) as plain Python, turned into the flat RtSceneDesc the C ABI
takes.  Defaults follow src/io.rs:116-125,132,152-153,159,178-180.
"""
import ctypes as C
import math

import numpy as np

from . import _abi


def lambertian(albedo):
    """Material::lambertian (src/material.rs:147)"""
    return (_abi.RT_MAT_LAMBERTIAN, tuple(albedo), 0.0, 0.0)


def glossy(albedo, reflectance=1.0, roughness=0.0):
    """Material::glossy (src/material.rs:153); scene-file order is (albedo, reflectance, roughness), src/io.rs:118-120"""
    return (_abi.RT_MAT_GLOSSY, tuple(albedo), float(reflectance), float(roughness))


def metal(albedo, roughness=0.0):
    """Material::metal (src/material.rs:161)"""
    return (_abi.RT_MAT_METAL, tuple(albedo), float(roughness), 0.0)


def dielectric(absorption, refractive_index=1.52, roughness=0.0):
    """Material::dielectric (src/material.rs:168)"""
    return (_abi.RT_MAT_DIELECTRIC, tuple(absorption), float(refractive_index), float(roughness))


def _cross_f32(a, b):
    """ultraviolet's Vec3::cross: each component is mul_add(p, q, -(r * s)) in float32."""
    a64, b64 = a.astype(np.float64), b.astype(np.float64)

    def comp(p, q, r, s):
        prod = np.float32(np.float32(r) * np.float32(s))  # rounded product
        return np.float32(p * q - np.float64(prod))  # fused: one rounding

    return np.array(
        [comp(a64[1], b64[2], a[2], b[1]), comp(a64[2], b64[0], a[0], b[2]), comp(a64[0], b64[1], a[1], b[0])],
        dtype=np.float32,
    )


class SceneDescription:
    """Collects objects in declaration order (which defines primitive ids) and builds an RtSceneDesc."""

    def __init__(self):
        self.materials = []  # (type, albedo, p0, p1)
        self.material_names = {}
        self.objects = []  # dicts
        self.meshes = []  # dicts with numpy arrays
        self.camera = dict(eye=(0, 0, 0), lookat=(0, 0, -1), vup=(0, 1, 0), fov=40.0, aperture=0.0, focus_dist=1.0)
        self.env = None  # float32 array [h, w, 3]
        self._keep = []

    # ---- materials -------------------------------------------------------
    def add_material(self, mat, name=None):
        self.materials.append(mat)
        idx = len(self.materials) - 1
        if name is not None:
            self.material_names[name] = idx
        return idx

    def _mat(self, m):
        if isinstance(m, str):
            return self.material_names[m]
        if isinstance(m, tuple):
            return self.add_material(m)
        return int(m)

    # ---- shapes (src/io.rs:128-187) ---------------------------------------
    def add_sphere(self, center, radius=1.0, material=0):
        self.objects.append(dict(type=_abi.RT_OBJ_SPHERE, material=self._mat(material), a=center, b=(radius, 0, 0)))

    def add_axis_box(self, center, size=(1.0, 1.0, 1.0), material=0):
        self.objects.append(dict(type=_abi.RT_OBJ_ABOX, material=self._mat(material), a=center, b=size))

    def add_box(self, center, size=(1.0, 1.0, 1.0), rotation=(0.0, 0.0, 0.0), material=0):
        """rotation in units of pi, as in the scene file"""
        self.objects.append(dict(type=_abi.RT_OBJ_CUBE, material=self._mat(material), a=center, b=size, c=rotation))

    def add_triangle(self, vertices, material=0, normals=None, two_sided=True):
        """Scene-file triangles are two-sided with the un-normalised face normal on all corners (src/io.rs:136-149)."""
        v = np.asarray(vertices, dtype=np.float32).reshape(3, 3)
        if normals is None:
            e1, e2 = v[1] - v[0], v[2] - v[0]
            n = _cross_f32(e1, e2)
            normals = np.stack([n, n, n])
        nn = np.asarray(normals, dtype=np.float32).reshape(3, 3)
        self.objects.append(
            dict(type=_abi.RT_OBJ_TRIANGLE, material=self._mat(material), tri_v=v.ravel(), tri_n=nn.ravel(), flag=int(two_sided))
        )

    def add_mesh(self, positions, normals, indices, material=0, translation=(0, 0, 0), scale=(1, 1, 1),
                 rotation=(0, 0, 0), cull_backface=False):
        """One tobj model.  `rotation` is in units of pi as in the scene file (src/io.rs:180 multiplies by pi)."""
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        if pos.shape != nrm.shape:
            raise ValueError("a mesh needs one normal per position (the reference indexes normals by vertex)")
        self.meshes.append(
            dict(pos=pos, nrm=nrm, idx=idx, translation=translation, scale=scale,
                 rotation=tuple(np.float32(r) * np.float32(math.pi) for r in rotation), cull=int(cull_backface))
        )
        self.objects.append(dict(type=_abi.RT_OBJ_MESH, material=self._mat(material), mesh=len(self.meshes) - 1))

    def set_camera(self, eye, lookat, fov, aperture=0.0, focus_dist=1.0, vup=(0.0, 1.0, 0.0)):
        self.camera = dict(eye=eye, lookat=lookat, vup=vup, fov=fov, aperture=aperture, focus_dist=focus_dist)

    def set_environment(self, rgb):
        """rgb: float32 [height, width, 3]; row 0 is straight up (src/render.rs:93-96)"""
        self.env = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.float32)

    def num_primitives(self):
        n = 0
        for o in self.objects:
            n += len(self.meshes[o["mesh"]]["idx"]) if o["type"] == _abi.RT_OBJ_MESH else 1
        return n

    # ---- flat description ---------------------------------------------------
    def to_desc(self):
        """Returns an RtSceneDesc; the arrays it points into stay alive as long as `self`."""
        mats = (_abi.RtMaterial * max(1, len(self.materials)))()
        for i, (t, alb, p0, p1) in enumerate(self.materials):
            mats[i].type = t
            mats[i].albedo = _abi.F3(*[float(x) for x in alb])
            mats[i].p0 = p0
            mats[i].p1 = p1
        objs = (_abi.RtObject * max(1, len(self.objects)))()
        for i, o in enumerate(self.objects):
            r = objs[i]
            r.type = o["type"]
            r.material = o["material"]
            r.a = _abi.F3(*[float(x) for x in o.get("a", (0, 0, 0))])
            r.b = _abi.F3(*[float(x) for x in o.get("b", (0, 0, 0))])
            r.c = _abi.F3(*[float(x) for x in o.get("c", (0, 0, 0))])
            if "tri_v" in o:
                r.tri_vertices = _abi.F9(*[float(x) for x in o["tri_v"]])
                r.tri_normals = _abi.F9(*[float(x) for x in o["tri_n"]])
            r.flag = o.get("flag", 0)
            r.mesh = o.get("mesh", -1)
        meshes = (_abi.RtMesh * max(1, len(self.meshes)))()
        for i, m in enumerate(self.meshes):
            r = meshes[i]
            r.positions = m["pos"].ctypes.data_as(C.POINTER(C.c_float))
            r.normals = m["nrm"].ctypes.data_as(C.POINTER(C.c_float))
            r.indices = m["idx"].ctypes.data_as(C.POINTER(C.c_uint32))
            r.n_vertices = len(m["pos"])
            r.n_faces = len(m["idx"])
            r.translation = _abi.F3(*[float(x) for x in m["translation"]])
            r.scale = _abi.F3(*[float(x) for x in m["scale"]])
            r.rotation = _abi.F3(*[float(x) for x in m["rotation"]])
            r.cull_backface = m["cull"]
        d = _abi.RtSceneDesc()
        d.materials = C.cast(mats, C.POINTER(_abi.RtMaterial))
        d.n_materials = len(self.materials)
        d.objects = C.cast(objs, C.POINTER(_abi.RtObject))
        d.n_objects = len(self.objects)
        d.meshes = C.cast(meshes, C.POINTER(_abi.RtMesh))
        d.n_meshes = len(self.meshes)
        cam = self.camera
        d.camera.eye = _abi.F3(*[float(x) for x in cam["eye"]])
        d.camera.lookat = _abi.F3(*[float(x) for x in cam["lookat"]])
        d.camera.vup = _abi.F3(*[float(x) for x in cam["vup"]])
        d.camera.fov_degrees = float(cam["fov"])
        d.camera.aperture = float(cam["aperture"])
        d.camera.focus_dist = float(cam["focus_dist"])
        if self.env is not None:
            d.env.rgb = self.env.ctypes.data_as(C.POINTER(C.c_float))
            d.env.height, d.env.width = self.env.shape[0], self.env.shape[1]
        self._keep = [mats, objs, meshes]
        return d


def to_ron(desc, mesh_paths=None):
    """Serialises a SceneDescription in the reference's scene-file schema (src/io.rs:24-94).
    Mesh objects need `mesh_paths[i]` = OBJ path of mesh i (one path per tobj model is not expressible, so
    consecutive objects sharing a path are written once)."""
    names = {i: n for n, i in desc.material_names.items()}

    def mname(i):
        return names.get(i, "m%d" % i)

    def t3(v):
        return "(%s)" % ", ".join(repr(float(np.float32(x))) for x in v)

    cam = desc.camera
    out = ["Scene(", "    hdr: None,",
           "    camera: (position: %s, lookat: %s, fov: %r, focus_dist: %r, apeture: %r)," % (
               t3(cam["eye"]), t3(cam["lookat"]), float(cam["fov"]), float(cam["focus_dist"]), float(cam["aperture"])),
           "    materials: {"]
    for i, (t, alb, p0, p1) in enumerate(desc.materials):
        if t == _abi.RT_MAT_LAMBERTIAN:
            body = "Lambertian(%s)" % t3(alb)
        elif t == _abi.RT_MAT_GLOSSY:
            body = "Glossy(%s, Some(%r), Some(%r))" % (t3(alb), p0, p1)
        elif t == _abi.RT_MAT_METAL:
            body = "Metal(%s, Some(%r))" % (t3(alb), p0)
        else:
            body = "Dielectric(%s, Some(%r), Some(%r))" % (t3(alb), p0, p1)
        out.append('        "%s": %s,' % (mname(i), body))
    out += ["    },", "    objects: ["]
    last_mesh_path = None
    for k, o in enumerate(desc.objects):
        t = o["type"]
        if t == _abi.RT_OBJ_SPHERE:
            shape = "Sphere(%s, Some(%r))" % (t3(o["a"]), float(np.float32(o["b"][0])))
        elif t == _abi.RT_OBJ_ABOX:
            shape = "AxisBox(%s, Some(%s))" % (t3(o["a"]), t3(o["b"]))
        elif t == _abi.RT_OBJ_CUBE:
            shape = "Box(%s, Some(%s), Some(%s))" % (t3(o["a"]), t3(o["b"]), t3(o["c"]))
        elif t == _abi.RT_OBJ_TRIANGLE:
            v = o["tri_v"]
            shape = "Triangle((%s))" % ", ".join("[%s]" % ", ".join(repr(float(x)) for x in v[3 * i:3 * i + 3]) for i in range(3))
        else:
            m = desc.meshes[o["mesh"]]
            path = mesh_paths[o["mesh"]]
            if path == last_mesh_path:
                continue
            last_mesh_path = path
            rot = [float(r) / math.pi for r in m["rotation"]]
            shape = 'Mesh("%s", Some(%s), Some(%s), Some(%s), %s)' % (
                path, t3(m["translation"]), t3(m["scale"]), t3(rot), "true" if m["cull"] else "false")
        out.append('        (name: Some("object %d"), shape: %s, material: "%s"),' % (k, shape, mname(o["material"])))
    out += ["    ],", ")"]
    return "\n".join(out) + "\n"
