"""Seeded scene generators for the configurations the core is measured on.

cornell()      the reference's Cornell-box scene (values of scene/example/cornell.ron)
rtiow()        the reference's default procedural scene (io::random_scene, src/io.rs:212-298,
               camera src/main.rs:174-182) with a seeded generator instead of fastrand
mesh_scene()   synthetic triangle-mesh scene: displaced spheres ("blobs") with vertex normals in a
               hollow diffuse room with an emissive panel; Lambertian / Glossy / Metal / Dielectric mix
random_rays()  ray batches for the closest-hit microbenchmark
sky_map()      procedural equirectangular HDR environment (sun disk + gradient)
"""
import math

import numpy as np

from .scene import SceneDescription, dielectric, glossy, lambertian, metal


def cornell():
    s = SceneDescription()
    s.set_camera(eye=(0.0, 0.0, -22.0), lookat=(0.0, 0.0, 0.0), fov=34.0, focus_dist=8.0, aperture=0.01)
    s.add_material(lambertian((0.73, 0.73, 0.73)), "diffuse")
    s.add_material(lambertian((15.0, 15.0, 15.0)), "light")
    s.add_material(lambertian((0.05, 0.05, 0.05)), "black")
    s.add_material(lambertian((0.65, 0.05, 0.05)), "red_diffuse")
    s.add_material(lambertian((0.12, 0.45, 0.15)), "green_diffuse")
    s.add_box((1.6, -2.0, 1.5), (3.0, 6.0, 3.0), (0.0, -0.10, 0.0), "diffuse")
    s.add_box((-1.8, -3.5, -1.8), (3.0, 3.0, 3.0), (0.0, 0.12, 0.0), "diffuse")
    s.add_axis_box((0.0, 0.0, -10.0), (-11.0, -11.0, -40.0), "black")  # hollow room
    s.add_axis_box((0.0, 5.0, 0.0), (10.0, 0.01, 10.0), "diffuse")  # ceiling
    s.add_axis_box((0.0, -5.0, 0.0), (10.0, 0.01, 10.0), "diffuse")  # floor
    s.add_axis_box((5.0, 0.0, 0.0), (0.01, 10.0, 10.0), "green_diffuse")
    s.add_axis_box((-5.0, 0.0, 0.0), (0.01, 10.0, 10.0), "red_diffuse")
    s.add_axis_box((0.0, 0.0, 5.0), (10.0, 10.0, 0.01), "diffuse")  # back wall
    s.add_axis_box((0.0, 4.99, 0.5), (2.8, 0.01, 2.8), "light")
    return s


def rtiow(seed=1):
    """random_scene(true, true, true, true, true) with the draw order of src/io.rs:221-283."""
    rng = np.random.Generator(np.random.PCG64(seed))

    def f32():
        return float(np.float32(rng.random(dtype=np.float32)))

    s = SceneDescription()
    s.set_camera(eye=(13.0, 2.0, 3.0), lookat=(0.0, 0.0, 0.0), fov=20.0, aperture=0.1, focus_dist=10.0)
    s.add_axis_box((-2.0, -0.5, -2.0), (50.0, 1.0, 50.0), glossy((0.55, 0.53, 0.56), 0.1, 0.7))
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose = f32()
            center = (np.float32(a) + np.float32(0.9) * np.float32(f32()), np.float32(0.2),
                      np.float32(b) + np.float32(0.9) * np.float32(f32()))
            d = math.sqrt((float(center[0]) - 4.0) ** 2 + (float(center[2]) - 0.0) ** 2)
            if d > 0.9:
                albedo = (f32() * f32(), f32() * f32(), f32() * f32())
                if choose < 0.3:
                    s.add_sphere(center, 0.2, glossy(albedo, f32() + 0.5, f32() * 0.5))
                elif choose < 0.6:
                    s.add_sphere(center, 0.2, lambertian(albedo))
                elif choose < 0.8:
                    s.add_sphere(center, 0.2, metal(albedo, 0.5 * f32()))
                elif choose < 0.9:
                    s.add_sphere(center, 0.2, lambertian((f32() * 6.0, f32() * 6.0, f32() * 6.0)))
                else:
                    s.add_sphere(center, 0.2, dielectric((f32(), f32(), f32()), 1.52, f32() * 0.5))
    s.add_sphere((4.0, 1.0, 0.0), 1.0, metal((0.7, 0.5, 0.3), 0.025))
    s.add_sphere((0.0, 1.0, 0.0), 1.0, dielectric((0.1, 0.1, 0.1), 1.52, 0.025))
    s.add_sphere((-4.0, 1.0, 0.0), 1.0, glossy((0.3, 0.2, 0.15), 0.6, 0.3))
    return s


def blob(rings, segments, seed):
    """A displaced unit sphere: positions [v,3], unit vertex normals [v,3], indices [f,3] (2*segments*(rings-1) faces)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    k1, k2 = rng.integers(2, 6, size=2)
    ph1, ph2 = rng.random(2) * 2.0 * math.pi
    amp = 0.08 + 0.12 * rng.random()
    theta = np.linspace(0.0, math.pi, rings + 1)[:, None]  # polar
    phi = np.linspace(0.0, 2.0 * math.pi, segments + 1)[None, :]
    r = 1.0 + amp * np.sin(k1 * theta + ph1) * np.cos(k2 * phi + ph2) * np.sin(theta)
    x = r * np.sin(theta) * np.cos(phi)
    y = r * np.cos(theta) * np.ones_like(phi)
    z = r * np.sin(theta) * np.sin(phi)
    pos = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    w = segments + 1
    faces = []
    for i in range(rings):
        a = i * w + np.arange(segments)
        b = a + 1
        c = a + w
        d = c + 1
        if i > 0:
            faces.append(np.stack([a, b, c], axis=-1))  # upper triangle (degenerate on the first ring)
        if i < rings - 1:
            faces.append(np.stack([b, d, c], axis=-1))  # lower triangle (degenerate on the last ring)
    idx = np.concatenate(faces).astype(np.uint32)
    # orient outward and build vertex normals from area-weighted face normals
    p0, p1, p2 = pos[idx[:, 0]], pos[idx[:, 1]], pos[idx[:, 2]]
    fn = np.cross(p1 - p0, p2 - p0)
    flip = np.einsum("ij,ij->i", fn, p0 + p1 + p2) < 0
    idx[flip] = idx[flip][:, [0, 2, 1]]
    fn[flip] = -fn[flip]
    vn = np.zeros_like(pos)
    for k in range(3):
        np.add.at(vn, idx[:, k], fn)
    # seam and pole duplicates share a position: give them the same normal
    vn2 = vn.reshape(rings + 1, w, 3)
    seam = vn2[:, 0] + vn2[:, -1]
    vn2[:, 0] = seam
    vn2[:, -1] = seam
    vn2[0, :] = vn2[0].sum(axis=0)
    vn2[-1, :] = vn2[-1].sum(axis=0)
    vn = vn2.reshape(-1, 3)
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-20)
    return pos.astype(np.float32), vn.astype(np.float32), idx


def mesh_scene(n_triangles=1_000_000, seed=7, rings=64, segments=64, n_variants=4):
    """About n_triangles triangles as a jittered 3-D grid of blobs inside a hollow room lit by a ceiling panel."""
    per_blob = 2 * segments * (rings - 1)
    n_blobs = max(1, int(round(n_triangles / per_blob)))
    side = max(1, int(math.ceil(n_blobs ** (1.0 / 3.0))))
    rng = np.random.Generator(np.random.PCG64(seed))
    variants = [blob(rings, segments, seed * 1000 + v) for v in range(n_variants)]
    s = SceneDescription()
    room = 24.0
    s.add_material(lambertian((0.7, 0.7, 0.7)), "room")
    s.add_material(lambertian((15.0, 15.0, 15.0)), "light")
    s.add_axis_box((0.0, 0.0, 0.0), (-room, -room, -room), "room")
    s.add_axis_box((0.0, room / 2 - 0.05, 0.0), (room * 0.6, 0.01, room * 0.6), "light")
    mats = [
        lambertian((0.75, 0.25, 0.2)),
        lambertian((0.2, 0.6, 0.3)),
        lambertian((0.25, 0.3, 0.75)),
        glossy((0.8, 0.7, 0.3), 0.9, 0.05),
        glossy((0.5, 0.3, 0.15), 0.3, 0.6),
        metal((0.8, 0.8, 0.85), 0.05),
        dielectric((0.3, 0.05, 0.1), 1.52, 0.01),
        dielectric((0.02, 0.02, 0.02), 1.33, 0.0),
    ]
    mat_ids = [s.add_material(m) for m in mats]
    extent = room * 0.7
    cell = extent / side
    count = 0
    for ix in range(side):
        for iy in range(side):
            for iz in range(side):
                if count >= n_blobs:
                    break
                c = (np.array([ix, iy, iz]) + 0.5) * cell - extent / 2 + (rng.random(3) - 0.5) * cell * 0.3
                c[1] -= room * 0.1
                scale = cell * (0.28 + 0.12 * rng.random())
                pos, nrm, idx = variants[count % n_variants]
                s.add_mesh(pos, nrm, idx, material=mat_ids[int(rng.integers(len(mat_ids)))],
                           translation=tuple(c), scale=(scale, scale, scale),
                           rotation=tuple(rng.random(3) * 2.0 - 1.0), cull_backface=False)
                count += 1
    s.set_camera(eye=(0.0, room * 0.05, -room * 0.47), lookat=(0.0, -room * 0.1, 0.0), fov=60.0, aperture=0.02,
                 focus_dist=room * 0.45)
    return s


def random_rays(n, bounds_min, bounds_max, seed=0x5EED, inflate=1.5):
    """Origins uniform in the bounding box inflated about its centre, directions uniform on the sphere: [n, 6] float32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lo, hi = np.asarray(bounds_min, dtype=np.float64), np.asarray(bounds_max, dtype=np.float64)
    c, h = (lo + hi) / 2, (hi - lo) / 2 * inflate
    o = c + (rng.random((n, 3)) * 2.0 - 1.0) * h
    z = rng.random(n) * 2.0 - 1.0
    ph = rng.random(n) * 2.0 * math.pi
    rr = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    d = np.stack([rr * np.cos(ph), rr * np.sin(ph), z], axis=-1)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def sky_map(width=2048, height=1024, sun_dir=(0.4, 0.7, -0.6), sun_radiance=50.0):
    """Equirectangular environment in the reference's convention (src/render.rs:89-96): column u = (atan2(x,z)+pi)/2pi,
    row 0 = straight up."""
    v = (np.arange(height) + 0.5) / height  # 0 = up
    u = (np.arange(width) + 0.5) / width
    polar = v[:, None] * math.pi
    az = u[None, :] * 2.0 * math.pi - math.pi
    dy = np.cos(polar) * np.ones_like(az)
    dx = np.sin(polar) * np.sin(az)
    dz = np.sin(polar) * np.cos(az)
    sd = np.asarray(sun_dir, dtype=np.float64)
    sd = sd / np.linalg.norm(sd)
    cosang = dx * sd[0] + dy * sd[1] + dz * sd[2]
    t = 0.5 * (dy + 1.0)
    sky = (1.0 - t)[..., None] * np.array([0.9, 0.85, 0.8]) + t[..., None] * np.array([0.25, 0.45, 0.9])
    ground = np.array([0.25, 0.22, 0.2])
    img = np.where((dy < 0)[..., None], ground, sky)
    img = img + (cosang > math.cos(math.radians(3.0)))[..., None] * sun_radiance
    return img.astype(np.float32)


def teaset(fixture=None, env_size=(2048, 1024)):
    """The reference's tea-set scene (values of scene/example/teapot.ron): four OBJ files = eight tobj models
    (17,228 triangles) on a table, ceramic / glass / steel, lit by an environment map.  The meshes come from
    the package's data/teaset.npz (made by tools/make_teaset_fixture.py from the reference's OBJ data); the HDR the scene
    names is not distributable, so a procedural sky stands in (sky_map)."""
    import os

    if fixture is None:
        fixture = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "teaset.npz")
    z = np.load(fixture)
    s = SceneDescription()
    s.set_camera(eye=(-0.5, 1.3, -3.0), lookat=(0.2, 1.0, 0.0), fov=50.0, focus_dist=3.0, aperture=0.1)
    s.add_material(metal((0.58, 0.56, 0.57), 0.1), "stainless steel")
    s.add_material(glossy((0.85, 0.82, 0.77), 0.95, 0.005), "ceramic")
    s.add_material(glossy((0.51, 0.29, 0.14), 0.25, 0.9), "wood")
    s.add_material(dielectric((0.6, 0.1, 0.25), 1.52, 0.025), "glass")
    s.add_axis_box((0.0, -0.005, 0.0), (6.5, 0.01, 5.0), "wood")

    def add(name, material, translation, scale, rotation, cull):
        for i in range(int(z[name + "_count"])):
            s.add_mesh(z["%s_%d_positions" % (name, i)], z["%s_%d_normals" % (name, i)], z["%s_%d_indices" % (name, i)],
                       material=material, translation=translation, scale=scale, rotation=rotation, cull_backface=cull)

    add("teapot", "ceramic", (-1.7, 1.25, 1.7), (0.5, 0.5, 0.5), (-0.2, 0.25, 0.0), False)
    add("teacup", "ceramic", (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (0.0, -0.7, 0.0), True)
    add("cup", "glass", (1.5, 0.6, -0.5), (0.05, 0.05, 0.05), (0.0, 0.0, 0.0), False)
    add("spoon", "stainless steel", (-0.9, 0.085, 0.3), (1.5, 1.5, 1.5), (0.025, -0.8, 0.0), True)
    if env_size:
        s.set_environment(sky_environment(env_size[0], env_size[1]))
    return s


def sky_environment(width, height):
    """The procedural sky as the renderer gets it from a scene file's `hdr:` entry: written as a Radiance .hdr and read
    back through the C++ host's RGBE reader (src/io.rs:106-111 reads the file with the radiant crate), i.e. with the
    values a real .hdr can hold: generated, written to a temporary file (run-length scanlines) and read back."""
    import os
    import tempfile

    from . import hostio

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sky.hdr")
        hostio.save_hdr(path, sky_map(width, height), rle=True)
        return hostio.load_hdr(path)
