// scene_io.hpp -- io::load_scene (src/io.rs:96-210) and the default procedural scene
// (io::random_scene src/io.rs:212-298 + make_default_setup src/main.rs:164-196) for the
// headless host: they fill the flat arrays rt_scene_create takes.
#ifndef RT_HOST_SCENE_IO_HPP
#define RT_HOST_SCENE_IO_HPP

#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtcore.h"
#include "image_io.hpp"
#include "obj.hpp"
#include "ron.hpp"

namespace sceneio {

struct SceneFile {
    std::vector<RtMaterial> materials;
    std::vector<RtObject> objects;
    std::vector<RtMesh> meshes;
    std::vector<std::unique_ptr<objio::Model>> models;  // storage the meshes point into
    RtCamera camera{};
    imageio::HdrImage env;
    std::vector<std::string> log;

    RtSceneDesc desc() const {
        RtSceneDesc d;
        std::memset(&d, 0, sizeof(d));
        d.materials = materials.data();
        d.n_materials = materials.size();
        d.objects = objects.data();
        d.n_objects = objects.size();
        d.meshes = meshes.data();
        d.n_meshes = meshes.size();
        d.camera = camera;
        if (!env.rgb.empty()) {
            d.env.rgb = env.rgb.data();
            d.env.width = env.width;
            d.env.height = env.height;
        }
        return d;
    }
};

namespace detail {
inline void vec3_of(const ron::Value *v, float out[3], const char *what) {
    if (!v || !(v->kind == ron::Value::Tuple || v->kind == ron::Value::List) || v->items.size() != 3)
        throw std::runtime_error(std::string("expected three numbers for ") + what);
    for (int k = 0; k < 3; k++) {
        if (v->items[k]->kind != ron::Value::Number) throw std::runtime_error(std::string("expected a number in ") + what);
        out[k] = (float)v->items[k]->number;
    }
}
inline float num_or(const ron::Value *v, float dflt) {
    if (!v) return dflt;
    const ron::Value *u = v->unwrap_option();
    if (!u) return dflt;
    if (u->kind != ron::Value::Number) throw std::runtime_error("expected a number");
    return (float)u->number;
}
inline void vec3_or(const ron::Value *v, float out[3], float dflt, const char *what) {
    const ron::Value *u = v ? v->unwrap_option() : nullptr;
    if (!u) {
        out[0] = out[1] = out[2] = dflt;
        return;
    }
    vec3_of(u, out, what);
}
inline RtMaterial material(int type, const float albedo[3], float p0, float p1) {
    RtMaterial m;
    m.type = type;
    m.albedo[0] = albedo[0], m.albedo[1] = albedo[1], m.albedo[2] = albedo[2];
    m.p0 = p0, m.p1 = p1;
    return m;
}
inline RtObject blank_object(int type, int material) {
    RtObject o;
    std::memset(&o, 0, sizeof(o));
    o.type = type;
    o.material = material;
    o.mesh = -1;
    return o;
}
inline bool try_open(const std::string &p) { return !p.empty() && std::ifstream(p).good(); }
}  // namespace detail

// Adds every model of an OBJ file as one mesh object, as src/io.rs:163-186 does.
inline void add_obj_meshes(SceneFile &s, const std::string &path, const float tr[3], const float sc[3],
                           const float rot_units_of_pi[3], bool cull_backface, int material) {
    std::vector<objio::Model> models = objio::load(path);
    if (models.empty()) throw std::runtime_error(path + ": no faces");
    for (auto &m : models) {
        if (m.normals.size() != m.positions.size())
            throw std::runtime_error(path + ": model '" + m.name + "' has no vertex normals (the reference indexes them unconditionally)");
        s.models.emplace_back(new objio::Model(std::move(m)));
        const objio::Model &mm = *s.models.back();
        RtMesh r;
        std::memset(&r, 0, sizeof(r));
        r.positions = mm.positions.data();
        r.normals = mm.normals.data();
        r.indices = mm.indices.data();
        r.n_vertices = mm.positions.size() / 3;
        r.n_faces = mm.indices.size() / 3;
        const float pi = 3.14159265358979323846f;
        for (int k = 0; k < 3; k++) {
            r.translation[k] = tr[k];
            r.scale[k] = sc[k];
            r.rotation[k] = rot_units_of_pi[k] * pi;  // src/io.rs:180
        }
        r.cull_backface = cull_backface ? 1 : 0;
        s.meshes.push_back(r);
        RtObject o = detail::blank_object(RT_OBJ_MESH, material);
        o.mesh = (int)s.meshes.size() - 1;
        s.objects.push_back(o);
    }
}

inline std::unique_ptr<SceneFile> load_scene(const std::string &path) {
    using namespace detail;
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open scene file " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    ron::ValuePtr doc = ron::parse(ss.str());
    if (doc->kind != ron::Value::Tuple) throw std::runtime_error("scene file: expected Scene(...)");
    std::unique_ptr<SceneFile> s(new SceneFile());

    // hdr: like the reference, a path that cannot be opened silently means "no environment map"
    if (const ron::Value *h = doc->field("hdr")) {
        const ron::Value *u = h->unwrap_option();
        if (u && u->kind == ron::Value::String) {
            std::string p = u->text, q = p;
            for (char &c : q)
                if (c == '\\') c = '/';
            std::string use = try_open(p) ? p : (try_open(q) ? q : std::string());
            if (!use.empty()) {
                try {
                    s->env = imageio::load_hdr(use);
                } catch (const std::exception &) {
                    s->env = imageio::HdrImage();  // radiant::load(...).ok()
                }
            }
        }
    }
    const ron::Value *cam = doc->field("camera");
    if (!cam) throw std::runtime_error("scene file: missing camera");
    vec3_of(cam->field("position"), s->camera.eye, "camera.position");
    vec3_of(cam->field("lookat"), s->camera.lookat, "camera.lookat");
    s->camera.vup[0] = 0.0f, s->camera.vup[1] = 1.0f, s->camera.vup[2] = 0.0f;  // src/io.rs:198
    s->camera.fov_degrees = num_or(cam->field("fov"), 40.0f);
    s->camera.focus_dist = num_or(cam->field("focus_dist"), 1.0f);
    s->camera.aperture = num_or(cam->field("apeture"), 0.0f);

    // materials: src/io.rs:115-126
    std::map<std::string, int> mat_index;
    const ron::Value *mats = doc->field("materials");
    if (!mats || mats->kind != ron::Value::Map) throw std::runtime_error("scene file: missing materials map");
    for (size_t i = 0; i < mats->items.size(); i++) {
        const ron::Value &k = *mats->keys[i], &v = *mats->items[i];
        if (k.kind != ron::Value::String || v.kind != ron::Value::Tuple || v.items.empty())
            throw std::runtime_error("scene file: bad material entry");
        float albedo[3];
        vec3_of(v.items[0].get(), albedo, "material albedo");
        auto arg = [&](size_t n) { return n < v.items.size() ? v.items[n].get() : nullptr; };
        RtMaterial m;
        if (v.text == "Lambertian")
            m = material(RT_MAT_LAMBERTIAN, albedo, 0.0f, 0.0f);
        else if (v.text == "Metal")
            m = material(RT_MAT_METAL, albedo, num_or(arg(1), 0.0f), 0.0f);
        else if (v.text == "Glossy")  // (albedo, reflectance, roughness): src/io.rs:118-120
            m = material(RT_MAT_GLOSSY, albedo, num_or(arg(1), 1.0f), num_or(arg(2), 0.0f));
        else if (v.text == "Dielectric")
            m = material(RT_MAT_DIELECTRIC, albedo, num_or(arg(1), 1.52f), num_or(arg(2), 0.0f));
        else
            throw std::runtime_error("scene file: unknown surface " + v.text);
        mat_index[k.text] = (int)s->materials.size();
        s->materials.push_back(m);
    }

    // objects: src/io.rs:128-187
    const ron::Value *objs = doc->field("objects");
    if (!objs || objs->kind != ron::Value::List) throw std::runtime_error("scene file: missing objects list");
    for (auto &op : objs->items) {
        const ron::Value *shape = op->field("shape"), *mname = op->field("material");
        if (!shape || shape->kind != ron::Value::Tuple || !mname || mname->kind != ron::Value::String)
            throw std::runtime_error("scene file: bad object");
        auto mi = mat_index.find(mname->text);
        if (mi == mat_index.end()) throw std::runtime_error("scene file: unknown material \"" + mname->text + "\"");
        int mat = mi->second;
        auto arg = [&](size_t n) { return n < shape->items.size() ? shape->items[n].get() : nullptr; };
        if (shape->text == "Sphere") {
            RtObject o = blank_object(RT_OBJ_SPHERE, mat);
            vec3_of(arg(0), o.a, "Sphere position");
            o.b[0] = num_or(arg(1), 1.0f);
            s->objects.push_back(o);
        } else if (shape->text == "AxisBox") {
            RtObject o = blank_object(RT_OBJ_ABOX, mat);
            vec3_of(arg(0), o.a, "AxisBox position");
            vec3_or(arg(1), o.b, 1.0f, "AxisBox size");
            s->objects.push_back(o);
        } else if (shape->text == "Box") {
            RtObject o = blank_object(RT_OBJ_CUBE, mat);
            vec3_of(arg(0), o.a, "Box position");
            vec3_or(arg(1), o.b, 1.0f, "Box size");
            vec3_or(arg(2), o.c, 0.0f, "Box rotation");
            s->objects.push_back(o);
        } else if (shape->text == "Triangle") {
            // Triangle(([x,y,z],[x,y,z],[x,y,z])): two-sided, face normal on every corner (src/io.rs:136-149)
            const ron::Value *t = arg(0);
            if (!t || t->items.size() != 3) throw std::runtime_error("scene file: Triangle needs three vertices");
            RtObject o = blank_object(RT_OBJ_TRIANGLE, mat);
            float v[3][3];
            for (int k = 0; k < 3; k++) vec3_of(t->items[k].get(), v[k], "Triangle vertex");
            float e1[3], e2[3];
            for (int k = 0; k < 3; k++) e1[k] = v[1][k] - v[0][k], e2[k] = v[2][k] - v[0][k];
            float n[3] = {std::fmaf(e1[1], e2[2], -e1[2] * e2[1]), std::fmaf(e1[2], e2[0], -e1[0] * e2[2]),
                          std::fmaf(e1[0], e2[1], -e1[1] * e2[0])};
            for (int k = 0; k < 3; k++)
                for (int c = 0; c < 3; c++) o.tri_vertices[3 * k + c] = v[k][c], o.tri_normals[3 * k + c] = n[c];
            o.flag = 1;
            s->objects.push_back(o);
        } else if (shape->text == "Mesh") {
            const ron::Value *loc = arg(0);
            if (!loc || loc->kind != ron::Value::String) throw std::runtime_error("scene file: Mesh needs a path");
            float tr[3], sc[3], rot[3];
            vec3_or(arg(1), tr, 0.0f, "Mesh translation");
            vec3_or(arg(2), sc, 1.0f, "Mesh scale");
            vec3_or(arg(3), rot, 0.0f, "Mesh rotation");
            const ron::Value *cull = arg(4);
            bool cull_backface = cull && cull->kind == ron::Value::Bool && cull->boolean;
            std::string p = loc->text, q = p;
            for (char &c : q)
                if (c == '\\') c = '/';
            add_obj_meshes(*s, try_open(p) ? p : q, tr, sc, rot, cull_backface, mat);
        } else {
            throw std::runtime_error("scene file: unknown shape " + shape->text);
        }
    }
    return s;
}

// Wyrand, the generator behind fastrand 1.9, seeded explicitly (the reference seeds it from the clock).
struct Wyrand {
    uint64_t state;
    explicit Wyrand(uint64_t seed) : state(seed) {}
    uint64_t next() {
        state += 0xa0761d6478bd642fULL;
        __uint128_t t = (__uint128_t)state * (state ^ 0xe7037ed1a0b428dbULL);
        return (uint64_t)(t >> 64) ^ (uint64_t)t;
    }
    float f32() { return (float)(uint32_t)(next() >> 41) * (1.0f / 8388608.0f); }  // 23 random mantissa bits
};

// io::random_scene(true, true, true, true, true) + the camera of make_default_setup
inline std::unique_ptr<SceneFile> default_scene(uint64_t seed) {
    using namespace detail;
    std::unique_ptr<SceneFile> s(new SceneFile());
    Wyrand rng(seed);
    auto add_mat = [&](RtMaterial m) {
        s->materials.push_back(m);
        return (int)s->materials.size() - 1;
    };
    auto sphere = [&](float x, float y, float z, float r, int mat) {
        RtObject o = blank_object(RT_OBJ_SPHERE, mat);
        o.a[0] = x, o.a[1] = y, o.a[2] = z;
        o.b[0] = r;
        s->objects.push_back(o);
    };
    {
        float a[3] = {0.55f, 0.53f, 0.56f};
        RtObject o = blank_object(RT_OBJ_ABOX, add_mat(material(RT_MAT_GLOSSY, a, 0.1f, 0.7f)));
        o.a[0] = -2.0f, o.a[1] = -0.5f, o.a[2] = -2.0f;
        o.b[0] = 50.0f, o.b[1] = 1.0f, o.b[2] = 50.0f;
        s->objects.push_back(o);
    }
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            float choose = rng.f32();
            float cx = (float)a + 0.9f * rng.f32(), cy = 0.2f, cz = (float)b + 0.9f * rng.f32();
            float dx = cx - 4.0f, dy = cy - 0.2f, dz = cz - 0.0f;
            if (std::sqrt(std::fmaf(dx, dx, std::fmaf(dy, dy, dz * dz))) > 0.9f) {
                float albedo[3];
                for (int k = 0; k < 3; k++) {
                    float p = rng.f32();
                    albedo[k] = p * rng.f32();
                }
                if (choose < 0.3f) {
                    float refl = rng.f32() + 0.5f, rough = rng.f32() * 0.5f;
                    sphere(cx, cy, cz, 0.2f, add_mat(material(RT_MAT_GLOSSY, albedo, refl, rough)));
                } else if (choose < 0.6f) {
                    sphere(cx, cy, cz, 0.2f, add_mat(material(RT_MAT_LAMBERTIAN, albedo, 0, 0)));
                } else if (choose < 0.8f) {
                    float fuzz = 0.5f * rng.f32();
                    sphere(cx, cy, cz, 0.2f, add_mat(material(RT_MAT_METAL, albedo, fuzz, 0)));
                } else if (choose < 0.9f) {
                    float l[3];
                    for (int k = 0; k < 3; k++) l[k] = rng.f32() * 6.0f;
                    sphere(cx, cy, cz, 0.2f, add_mat(material(RT_MAT_LAMBERTIAN, l, 0, 0)));
                } else {
                    float ab[3];
                    for (int k = 0; k < 3; k++) ab[k] = rng.f32();
                    float rough = rng.f32() * 0.5f;
                    sphere(cx, cy, cz, 0.2f, add_mat(material(RT_MAT_DIELECTRIC, ab, 1.52f, rough)));
                }
            }
        }
    }
    float steel[3] = {0.7f, 0.5f, 0.3f}, glass[3] = {0.1f, 0.1f, 0.1f}, gloss[3] = {0.3f, 0.2f, 0.15f};
    sphere(4.0f, 1.0f, 0.0f, 1.0f, add_mat(material(RT_MAT_METAL, steel, 0.025f, 0)));
    sphere(0.0f, 1.0f, 0.0f, 1.0f, add_mat(material(RT_MAT_DIELECTRIC, glass, 1.52f, 0.025f)));
    sphere(-4.0f, 1.0f, 0.0f, 1.0f, add_mat(material(RT_MAT_GLOSSY, gloss, 0.6f, 0.3f)));
    // src/main.rs:174-182
    s->camera.eye[0] = 13.0f, s->camera.eye[1] = 2.0f, s->camera.eye[2] = 3.0f;
    s->camera.lookat[0] = s->camera.lookat[1] = s->camera.lookat[2] = 0.0f;
    s->camera.vup[1] = 1.0f;
    s->camera.fov_degrees = 20.0f;
    s->camera.aperture = 0.1f;
    s->camera.focus_dist = 10.0f;
    return s;
}

}  // namespace sceneio
#endif
