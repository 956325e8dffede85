/*
 * oracle.cpp -- TEST INFRASTRUCTURE.  CPU restatement of the hot path of
 * Teetlez/Rust-Raytracer, used only as the checker in tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  The product
 * (librtcore) never includes, links or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
 * this path and cannot be built here (no Rust toolchain, no crates on disk), so
 * this restatement is pinned only by known answers derived by hand from the
 * cited source lines (tests/golden/kat.json).
 *
 * Shape: it keeps the reference's own structure on purpose -- a pointer-linked
 * binary tree with one primitive per leaf, recursive right-then-left traversal
 * with t_max narrowing, virtual dispatch per leaf, float32 everywhere, the same
 * inequalities and epsilons.  Every function names the reference lines it
 * follows (paths relative to the reference repository root).
 *
 * Third-party arithmetic that is not in the reference tree (crates.io, Cargo.toml
 * caret requirements, no lock file): ultraviolet ^0.9.1 (Vec3 / Rotor3),
 * fastrand ^1.9.0, quasirandom ^0.3.0.  Their published algorithms are restated
 * here: Vec3::dot / cross / mag_sq are fused multiply-add chains, normalized()
 * multiplies by 1/mag, Rotor3::from_euler_angles(roll, pitch, yaw) =
 * rot_xz(yaw) * rot_yz(pitch) * rot_xy(roll).  The reference's generators are
 * unseeded (thread-local Wyrand, R2 sequence with a random start), so only their
 * distributions can be kept: uniform draws are 23-bit multiples of 2^-23 in
 * [0,1) from a counter-based hash of (seed, pixel, sample, bounce, dimension),
 * and the pixel jitter is an R2 additive recurrence with a hashed start per
 * (pixel, pass).  sin/cos/acos/atan2/exp come from det_math.h.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off -mfma: the only fused
 * operations are the explicit fmaf calls that mirror ultraviolet's mul_add).
 */
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../include/rtcore.h"
#include "det_math.h"

namespace {

// ---------------------------------------------------------------------------
// f32 helpers with Rust semantics
// ---------------------------------------------------------------------------
// f32::min / f32::max ignore a NaN operand.
inline float rmin(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
inline float rmax(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
// f32::clamp keeps NaN.
inline float rclamp(float x, float lo, float hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
inline float recip(float x) { return 1.0f / x; }
inline float powi5(float x) {
    float x2 = x * x;
    return (x2 * x2) * x;
}

const float PI = 3.14159265358979323846f;
const float F32_EPSILON = 1.1920929e-7f;

// ultraviolet::Vec3
struct Vec3 {
    float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Vec3 operator+(Vec3 o) const { return Vec3(x + o.x, y + o.y, z + o.z); }
    Vec3 operator-(Vec3 o) const { return Vec3(x - o.x, y - o.y, z - o.z); }
    Vec3 operator*(Vec3 o) const { return Vec3(x * o.x, y * o.y, z * o.z); }
    Vec3 operator/(Vec3 o) const { return Vec3(x / o.x, y / o.y, z / o.z); }
    Vec3 operator*(float s) const { return Vec3(x * s, y * s, z * s); }
    Vec3 operator/(float s) const { return Vec3(x / s, y / s, z / s); }
    Vec3 operator-() const { return Vec3(-x, -y, -z); }
    float dot(Vec3 o) const { return std::fmaf(x, o.x, std::fmaf(y, o.y, z * o.z)); }
    Vec3 cross(Vec3 o) const {
        return Vec3(std::fmaf(y, o.z, -z * o.y), std::fmaf(z, o.x, -x * o.z), std::fmaf(x, o.y, -y * o.x));
    }
    float mag_sq() const { return std::fmaf(x, x, std::fmaf(y, y, z * z)); }
    float mag() const { return std::sqrt(mag_sq()); }
    Vec3 normalized() const {
        float r = 1.0f / mag();
        return Vec3(x * r, y * r, z * r);
    }
    Vec3 reflected(Vec3 n) const { return *this - (n * (2.0f * dot(n))); }
    Vec3 refracted(Vec3 n, float eta) const {
        float ndi = n.dot(*this);
        float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
        if (k < 0.0f) return Vec3(0, 0, 0);
        return (*this) * eta - n * (eta * ndi + std::sqrt(k));
    }
    float component_max() const { return rmax(rmax(x, y), z); }
    float component_min() const { return rmin(rmin(x, y), z); }
    Vec3 min_by_component(Vec3 o) const { return Vec3(rmin(x, o.x), rmin(y, o.y), rmin(z, o.z)); }
    Vec3 max_by_component(Vec3 o) const { return Vec3(rmax(x, o.x), rmax(y, o.y), rmax(z, o.z)); }
    Vec3 clamped(Vec3 lo, Vec3 hi) const {
        return Vec3(rmin(rmax(x, lo.x), hi.x), rmin(rmax(y, lo.y), hi.y), rmin(rmax(z, lo.z), hi.z));
    }
};
inline Vec3 operator*(float s, Vec3 v) { return Vec3(s * v.x, s * v.y, s * v.z); }
inline Vec3 splat(float s) { return Vec3(s, s, s); }

// ultraviolet::Rotor3 {s, bv:{xy,xz,yz}}
struct Rotor3 {
    float s, xy, xz, yz;
    static Rotor3 from_angle_plane(float angle, float pxy, float pxz, float pyz) {
        float half = angle * 0.5f;
        float sn = std::sin(half), cs = std::cos(half);
        return Rotor3{cs, pxy * -sn, pxz * -sn, pyz * -sn};
    }
    Rotor3 mul(const Rotor3 &q) const {
        Rotor3 r;
        r.s = s * q.s - xy * q.xy - xz * q.xz - yz * q.yz;
        r.xy = xy * q.s + s * q.xy + yz * q.xz - xz * q.yz;
        r.xz = xz * q.s + s * q.xz - yz * q.xy + xy * q.yz;
        r.yz = yz * q.s + s * q.yz + xz * q.xy - xy * q.xz;
        return r;
    }
    static Rotor3 from_euler_angles(float roll, float pitch, float yaw) {
        Rotor3 rxz = from_angle_plane(yaw, 0, 1, 0);
        Rotor3 ryz = from_angle_plane(pitch, 0, 0, 1);
        Rotor3 rxy = from_angle_plane(roll, 1, 0, 0);
        return rxz.mul(ryz).mul(rxy);
    }
    Rotor3 normalized() const {
        float m2 = s * s + std::fmaf(xy, xy, std::fmaf(xz, xz, yz * yz));
        float r = 1.0f / std::sqrt(m2);
        return Rotor3{s * r, xy * r, xz * r, yz * r};
    }
    Rotor3 reversed() const { return Rotor3{s, -xy, -xz, -yz}; }
    Vec3 rotate(Vec3 v) const {
        float fx = s * v.x + xy * v.y + xz * v.z;
        float fy = s * v.y - xy * v.x + yz * v.z;
        float fz = s * v.z - xz * v.x - yz * v.y;
        float fw = xy * v.z - xz * v.y + yz * v.x;
        return Vec3(s * fx + xy * fy + xz * fz + yz * fw, s * fy - xy * fx - xz * fw + yz * fz,
                    s * fz + xy * fw - xz * fx - yz * fy);
    }
};

// src/ray.rs:3-21
struct Ray {
    Vec3 pos, dir;
    Ray() {}
    Ray(Vec3 p, Vec3 d) : pos(p), dir(d.normalized()) {}
    Vec3 at(float t) const { return pos + t * dir; }
};

// src/ray.rs:23-51
struct Onb {
    Vec3 u, v, w;
    static Onb from_w(Vec3 n) {
        Onb o;
        o.w = n.normalized();
        Vec3 a = (std::fabs(o.w.x) > 0.9f) ? Vec3(0, 1, 0) : Vec3(1, 0, 0);
        o.v = o.w.cross(a).normalized();
        o.u = o.w.cross(o.v);
        return o;
    }
    Vec3 local(Vec3 a) const { return a.x * u + a.y * v + a.z * w; }
};

// ---------------------------------------------------------------------------
// counter-based random numbers (replace fastrand / quasirandom)
// ---------------------------------------------------------------------------
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
inline uint32_t path_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t k = mix32(seed ^ 0xA511E9B3U);
    k = mix32(k ^ pixel);
    k = mix32(k + sample * 0x9E3779B9U);
    return k;
}
inline float u01(uint32_t bits) { return (float)(bits >> 9) * (1.0f / 8388608.0f); }
// draw number `dim` of bounce `bounce` on a path
inline float draw(uint32_t key, uint32_t bounce, uint32_t dim) {
    uint32_t ctr = bounce * 8U + dim;
    return u01(mix32(key ^ mix32(ctr * 0x9E3779B9U + 0x7F4A7C15U)));
}
enum { DIM_R1 = 0, DIM_R2 = 1, DIM_S0 = 2, DIM_S1 = 3, DIM_S2 = 4, DIM_CHOICE = 5, DIM_RR = 6 };
// R2 jitter: start hashed per (pixel, pass), then the additive recurrence in 32-bit fixed point
inline void pixel_jitter(uint32_t seed, uint32_t pixel, uint32_t pass, uint32_t s, float *jx, float *jy) {
    uint32_t pk = mix32(mix32(mix32(seed ^ 0x243F6A88U) ^ pixel) + pass * 0x9E3779B9U);
    uint32_t sx = mix32(pk + 0x68E31DA4U), sy = mix32(pk + 0xB5297A4DU);
    uint32_t x = sx + (s + 1U) * 0xC13FA9A9U; /* 2^32 / g,   g = plastic number */
    uint32_t y = sy + (s + 1U) * 0x91E10DA5U; /* 2^32 / g^2 */
    *jx = u01(x);
    *jy = u01(y);
}

// ---------------------------------------------------------------------------
// materials: src/material.rs
// ---------------------------------------------------------------------------
const float AIR_INDEX = 1.00028f;  // src/material.rs:9

struct Material {
    int type;
    Vec3 albedo;
    float p0, p1;
};

struct HitRecord {
    float t;
    Vec3 point, normal;
    const Material *material;
    int32_t prim_id;
};

struct Scatter {
    Vec3 attenuation;
    Ray ray;
};

// src/random.rs:14-22
inline Vec3 in_concentric_disk(float r1, float r2) {
    float a = (2.0f * r1) - 1.0f, b = (2.0f * r2) - 1.0f;
    float radius, phi;
    if ((a * a) > (b * b)) {
        radius = a;
        phi = (PI / 4.0f) * (b / a);
    } else {
        radius = b;
        phi = (PI / 2.0f) - ((PI / 4.0f) * (a / b));
    }
    float sn, cs;
    detm::sincos(phi, &sn, &cs);
    return Vec3(cs * radius, sn * radius, 0.0f);
}
// src/random.rs:25-32
inline Vec3 on_cosine_sphere(float r1, float r2) {
    float z = std::sqrt(1.0f - r2);
    float phi = 2.0f * PI * r1;
    float sn, cs;
    detm::sincos(phi, &sn, &cs);
    float x = cs * std::sqrt(r2);
    float y = sn * std::sqrt(r2);
    return Vec3(x, y, z);
}
// src/random.rs:35-43
inline Vec3 on_uniform_sphere(float r1, float r2) {
    float phi = 2.0f * PI * r1;
    float theta = detm::acos((2.0f * r2) - 1.0f);
    float sp, cp, st, ct;
    detm::sincos(phi, &sp, &cp);
    detm::sincos(theta, &st, &ct);
    return Vec3(cp * st, sp * st, ct);
}
// src/random.rs:8-11: radius scales linearly with the third draw
inline Vec3 in_unit_sphere(float r1, float r2, float u) { return on_uniform_sphere(r1, r2) * u; }
// src/random.rs:46-48
inline Vec3 on_hemisphere(Vec3 normal, float r1, float r2) {
    return Onb::from_w(normal).local(on_cosine_sphere(r1, r2));
}

// src/material.rs:90-94
inline float schlick(float cosine, float ni, float nt) {
    float q = (ni - nt) / (ni + nt);
    float r0 = q * q;
    return rclamp(r0 + (1.0f - r0) * powi5(1.0f - cosine), 0.0f, 1.0f);
}

// src/material.rs:176-183 dispatching to :28-35, :44-59, :67-81, :96-136
Scatter scatter(const Material &m, const Ray &ray, const HitRecord &hit, uint32_t key, uint32_t bounce) {
    float r1 = draw(key, bounce, DIM_R1), r2 = draw(key, bounce, DIM_R2);
    Scatter out;
    switch (m.type) {
        case RT_MAT_LAMBERTIAN: {
            Vec3 direction = on_hemisphere(hit.normal, r1, r2);
            out.attenuation = m.albedo;
            out.ray = Ray(hit.point, direction);
            return out;
        }
        case RT_MAT_GLOSSY: {
            float reflectance = m.p0, roughness = m.p1;
            Vec3 rnd = in_unit_sphere(draw(key, bounce, DIM_S0), draw(key, bounce, DIM_S1), draw(key, bounce, DIM_S2));
            Vec3 jn = (hit.normal + (rnd * roughness)).normalized();
            float cosine = (-ray.dir).dot(jn);
            float prob = schlick(cosine, AIR_INDEX, 1.0f + reflectance);
            Vec3 color, out_dir;
            if (draw(key, bounce, DIM_CHOICE) <= prob) {
                color = splat(1.0f) * 0.9f;
                out_dir = ray.dir.reflected(jn);
            } else {
                color = m.albedo;
                out_dir = on_hemisphere(hit.normal, r1, r2);
            }
            out.attenuation = color;
            out.ray = Ray(hit.point, out_dir);
            return out;
        }
        case RT_MAT_METAL: {
            float roughness = m.p0;
            Vec3 rnd = in_unit_sphere(draw(key, bounce, DIM_S0), draw(key, bounce, DIM_S1), draw(key, bounce, DIM_S2));
            Vec3 jn = (hit.normal + (rnd * roughness)).normalized();
            Vec3 out_dir = ray.dir.reflected(jn);
            float cosine = (-ray.dir).dot(jn);
            out.attenuation =
                (m.albedo + (splat(1.0f) - m.albedo) * powi5(1.0f - cosine)).clamped(splat(0.0f), splat(1.0f));
            out.ray = Ray(hit.point, out_dir);
            return out;
        }
        default: {  // RT_MAT_DIELECTRIC
            float ior = m.p0, roughness = m.p1;
            Vec3 rnd = in_unit_sphere(r1, r2, draw(key, bounce, DIM_S0));
            Vec3 jn = (hit.normal + (rnd * roughness)).normalized();
            Vec3 outward, color;
            float ni, nt, cosine;
            if (ray.dir.dot(hit.normal) > 0.0f) {
                Vec3 absorbance = m.albedo * -hit.t * 2.0f;
                outward = -jn;
                ni = ior;
                nt = AIR_INDEX;
                cosine = ray.dir.dot(jn);
                color = Vec3(detm::exp(absorbance.x), detm::exp(absorbance.y), detm::exp(absorbance.z));
            } else {
                outward = jn;
                ni = AIR_INDEX;
                nt = ior;
                cosine = (-ray.dir).dot(jn);
                color = splat(1.0f) * 0.9f;
            }
            out.attenuation = color;
            if ((ni / nt) * std::sqrt(1.0f - (cosine * cosine)) <= 1.0f) {
                float prob = schlick(cosine, ni, nt);
                Vec3 out_dir = (draw(key, bounce, DIM_CHOICE) <= prob) ? ray.dir.reflected(outward)
                                                                       : ray.dir.refracted(outward, ni / nt);
                out.ray = Ray(hit.point, out_dir);
            } else {
                out.ray = Ray(hit.point, ray.dir.reflected(outward));
            }
            return out;
        }
    }
}

// ---------------------------------------------------------------------------
// geometry: src/tracer/*.rs
// ---------------------------------------------------------------------------
// src/tracer/cube.rs:151-199
struct Aabb {
    Vec3 min, max;
    bool hit(Vec3 origin, Vec3 inv_d, float t_min, float t_max) const {
        Vec3 t0 = (min - origin) * inv_d;
        Vec3 t1 = (max - origin) * inv_d;
        Vec3 t_near = t0.min_by_component(t1);
        Vec3 t_far = t0.max_by_component(t1);
        return rmax(t_near.component_max(), t_min) <= rmin(t_far.component_min(), t_max);
    }
    Vec3 center() const { return (max + min) * 0.5f; }
    static Aabb surrounding_box(const Aabb &a, const Aabb &b) {
        Aabb r;
        r.min = Vec3(rmin(a.min.x, b.min.x) - F32_EPSILON, rmin(a.min.y, b.min.y) - F32_EPSILON,
                     rmin(a.min.z, b.min.z) - F32_EPSILON);
        r.max = Vec3(rmax(a.max.x, b.max.x) + F32_EPSILON, rmax(a.max.y, b.max.y) + F32_EPSILON,
                     rmax(a.max.z, b.max.z) + F32_EPSILON);
        return r;
    }
    bool surrounds_axis(const Aabb &o, int axis) const { return min[axis] < o.min[axis] && max[axis] > o.max[axis]; }
};

struct Counters {
    uint64_t rays = 0, node_visits = 0, prim_tests = 0;
};
thread_local Counters *tl_counters = nullptr;
// per-node and per-primitive counts are optional (oracle_set_counting): a timed run leaves them out
thread_local Counters *tl_detail = nullptr;
std::atomic<bool> g_count_detail(true);

// src/tracer/hittable.rs:7-11
struct Hittable {
    virtual ~Hittable() {}
    virtual bool hit(const Ray &ray, float t_min, float t_max, HitRecord &rec) const = 0;
    virtual Aabb bounding_box() const = 0;
    // leaves in the order the reference's traversal reaches them (right child first)
    virtual void visit_order(std::vector<int32_t> &out) const = 0;
};

// src/tracer/sphere.rs
struct Sphere : Hittable {
    Vec3 center;
    float radius;
    const Material *material;
    int32_t prim_id;
    bool hit(const Ray &ray, float t_min, float t_max, HitRecord &rec) const override {
        if (tl_detail) tl_detail->prim_tests++;
        Vec3 oc = ray.pos - center;
        float half_b = oc.dot(ray.dir);
        float disc = half_b * half_b - (oc.mag_sq() - (radius * radius));
        if (disc > 0.0f) {
            float h = std::sqrt(disc);
            float temp = -half_b - h;
            if (temp < t_max && temp > t_min) {
                Vec3 p = ray.at(temp);
                rec = HitRecord{temp, p, (p - center).normalized(), material, prim_id};
                return true;
            }
            temp = -half_b + h;
            if (temp < t_max && temp > t_min) {
                Vec3 p = ray.at(temp);
                rec = HitRecord{temp, p, (p - center).normalized(), material, prim_id};
                return true;
            }
        }
        return false;
    }
    Aabb bounding_box() const override {
        return Aabb{center - splat(1.0f) * std::fabs(radius), center + splat(1.0f) * std::fabs(radius)};
    }
    void visit_order(std::vector<int32_t> &out) const override { out.push_back(prim_id); }
};

const float TOLERANCE = 0.0001f;  // src/tracer/cube.rs:17

// src/tracer/cube.rs:9-80
struct ABox : Hittable {
    Vec3 min, max;
    bool hollow;
    const Material *material;
    int32_t prim_id;
    ABox() {}
    ABox(const float c[3], const float s[3], const Material *m, int32_t id) : material(m), prim_id(id) {
        hollow = rmin(rmin(s[0], s[1]), s[2]) < 0.0f;
        min = Vec3(c[0] - std::fabs(s[0] * 0.5f), c[1] - std::fabs(s[1] * 0.5f), c[2] - std::fabs(s[2] * 0.5f));
        max = Vec3(c[0] + std::fabs(s[0] * 0.5f), c[1] + std::fabs(s[1] * 0.5f), c[2] + std::fabs(s[2] * 0.5f));
    }
    bool hit_inner(const Ray &ray, float t_min, float t_max, HitRecord &rec) const {
        Vec3 inv_d(recip(ray.dir.x), recip(ray.dir.y), recip(ray.dir.z));
        Vec3 t0 = (min - ray.pos) * inv_d;
        Vec3 t1 = (max - ray.pos) * inv_d;
        float t_near = rmax(t0.min_by_component(t1).component_max(), t_min);
        float t_far = rmin(t0.max_by_component(t1).component_min(), t_max);
        if (t_near <= t_far) {
            float t;
            if (t_near > t_min)
                t = t_near;
            else if (t_far < t_max)
                t = t_far;
            else
                return false;
            Vec3 p = ray.at(t);
            Vec3 n;
            if (std::fabs(p.x - max.x) < TOLERANCE)
                n = Vec3(1, 0, 0);
            else if (std::fabs(p.y - max.y) < TOLERANCE)
                n = Vec3(0, 1, 0);
            else if (std::fabs(p.z - max.z) < TOLERANCE)
                n = Vec3(0, 0, 1);
            else if (std::fabs(p.x - min.x) < TOLERANCE)
                n = -Vec3(1, 0, 0);
            else if (std::fabs(p.y - min.y) < TOLERANCE)
                n = -Vec3(0, 1, 0);
            else if (std::fabs(p.z - min.z) < TOLERANCE)
                n = -Vec3(0, 0, 1);
            else
                n = Vec3(0, 0, 0);
            n = n * (hollow ? -1.0f : 1.0f);
            rec = HitRecord{t, p, n, material, prim_id};
            return true;
        }
        return false;
    }
    bool hit(const Ray &ray, float t_min, float t_max, HitRecord &rec) const override {
        if (tl_detail) tl_detail->prim_tests++;
        return hit_inner(ray, t_min, t_max, rec);
    }
    Aabb bounding_box() const override { return Aabb{min, max}; }
    void visit_order(std::vector<int32_t> &out) const override { out.push_back(prim_id); }
};

// src/tracer/cube.rs:82-149
struct Cube : Hittable {
    ABox axis_box;
    Vec3 center;
    Rotor3 rotation;
    Cube(const float c[3], const float s[3], const float rot[3], const Material *m, int32_t id)
        : axis_box(c, s, m, id), center(c[0], c[1], c[2]) {
        rotation = Rotor3::from_euler_angles(rot[2] * PI, rot[0] * PI, rot[1] * PI).normalized();
    }
    bool hit(const Ray &ray, float t_min, float t_max, HitRecord &rec) const override {
        if (tl_detail) tl_detail->prim_tests++;
        Vec3 rot_pos = rotation.reversed().rotate(ray.pos - center) + center;
        Vec3 rot_dir = rotation.reversed().rotate(ray.dir);
        HitRecord h;
        if (!axis_box.hit_inner(Ray(rot_pos, rot_dir), t_min, t_max, h)) return false;
        rec = HitRecord{h.t, ray.at(h.t), rotation.rotate(h.normal), h.material, h.prim_id};
        return true;
    }
    Aabb bounding_box() const override {
        Aabb b = axis_box.bounding_box();
        Vec3 corners[8] = {Vec3(b.min.x, b.min.y, b.min.z), Vec3(b.max.x, b.min.y, b.min.z),
                           Vec3(b.min.x, b.max.y, b.min.z), Vec3(b.max.x, b.max.y, b.min.z),
                           Vec3(b.min.x, b.min.y, b.max.z), Vec3(b.max.x, b.min.y, b.max.z),
                           Vec3(b.min.x, b.max.y, b.max.z), Vec3(b.max.x, b.max.y, b.max.z)};
        Vec3 mn(INFINITY, INFINITY, INFINITY), mx(-INFINITY, -INFINITY, -INFINITY);
        for (auto &c : corners) {
            Vec3 v = rotation.rotate(c - center) + center;
            mn = Vec3(rmin(mn.x, v.x), rmin(mn.y, v.y), rmin(mn.z, v.z));
            mx = Vec3(rmax(mx.x, v.x), rmax(mx.y, v.y), rmax(mx.z, v.z));
        }
        return Aabb{mn, mx};
    }
    void visit_order(std::vector<int32_t> &out) const override { out.push_back(axis_box.prim_id); }
};

// src/tracer/triangle.rs
struct Triangle : Hittable {
    Vec3 vertices[3];
    Vec3 normals[3];
    const Material *material;
    bool two_sided;
    int32_t prim_id;
    bool hit(const Ray &ray, float t_min, float t_max, HitRecord &rec) const override {
        if (tl_detail) tl_detail->prim_tests++;
        Vec3 edge1 = vertices[1] - vertices[0];
        Vec3 edge2 = vertices[2] - vertices[0];
        Vec3 h = ray.dir.cross(edge2);
        float a = edge1.dot(h);
        if ((!two_sided && std::signbit(a)) || std::fabs(a) < 1e-6f) return false;
        float f = recip(a);
        Vec3 s = ray.pos - vertices[0];
        float u = f * s.dot(h);
        if (!(u >= 0.0f && u <= 1.0f)) return false;
        Vec3 q = s.cross(edge1);
        float v = f * ray.dir.dot(q);
        if (v < 0.0f || u + v > 1.0f) return false;
        float t = f * edge2.dot(q);
        if (t < t_min || t > t_max) return false;
        Vec3 normal = ((1.0f - (u + v)) * normals[0] + u * normals[1] + v * normals[2]).normalized();
        rec = HitRecord{t, ray.at(t), normal, material, prim_id};
        return true;
    }
    Aabb bounding_box() const override {
        return Aabb{Vec3(rmin(vertices[0].x, rmin(vertices[1].x, vertices[2].x)),
                         rmin(vertices[0].y, rmin(vertices[1].y, vertices[2].y)),
                         rmin(vertices[0].z, rmin(vertices[1].z, vertices[2].z))),
                    Vec3(rmax(vertices[0].x, rmax(vertices[1].x, vertices[2].x)),
                         rmax(vertices[0].y, rmax(vertices[1].y, vertices[2].y)),
                         rmax(vertices[0].z, rmax(vertices[1].z, vertices[2].z)))};
    }
    void visit_order(std::vector<int32_t> &out) const override { out.push_back(prim_id); }
};

// Stable top-down merge sort.  Rust's slice::sort_by is a stable merge sort too, but
// box_compare (src/tracer/bvh.rs:120-139) is not a strict weak order, so the exact
// permutation depends on the algorithm; this one is the build's definition.
struct SortItem {
    Hittable *obj;
    Aabb box;
};

// src/tracer/bvh.rs:120-139 -> true when a orders strictly before b
inline int box_compare(const Aabb &a, const Aabb &b, int axis) {
    Vec3 ca = a.max + a.min * 0.5f;  // sic
    Vec3 cb = b.max + b.min * 0.5f;
    if (a.surrounds_axis(b, axis)) return -1;
    if (b.surrounds_axis(a, axis)) return 1;
    if (ca[axis] < cb[axis]) return -1;
    if (ca[axis] > cb[axis]) return 1;
    return 0;
}

void merge_sort(SortItem *a, SortItem *tmp, size_t n, int axis) {
    if (n < 2) return;
    size_t mid = n / 2;
    merge_sort(a, tmp, mid, axis);
    merge_sort(a + mid, tmp, n - mid, axis);
    size_t i = 0, j = mid, k = 0;
    while (i < mid && j < n) {
        if (box_compare(a[j].box, a[i].box, axis) < 0)
            tmp[k++] = a[j++];
        else
            tmp[k++] = a[i++];
    }
    while (i < mid) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    std::memcpy(a, tmp, n * sizeof(SortItem));
}

// src/tracer/bvh.rs:31-167
struct Bvh : Hittable {
    Aabb aabb_box;
    Hittable *left = nullptr, *right = nullptr;  // Branch(Bvh) or Leaf(primitive / mesh)
    std::vector<std::unique_ptr<Bvh>> owned;

    static int largest_axis(const SortItem *objs, size_t n) {
        Aabb bbox = objs[0].box;
        for (size_t i = 0; i < n; i++) bbox = Aabb::surrounding_box(bbox, objs[i].box);
        float width = bbox.max.x - bbox.min.x;
        float height = bbox.max.y - bbox.min.y;
        float depth = bbox.max.z - bbox.min.z;
        if (width > height && width > depth) return 0;
        if (height > depth) return 1;
        return 2;
    }
    static size_t true_middle(const SortItem *objs, size_t n, int axis) {
        size_t mid = 1;
        float mid_world = (objs[0].box.min[axis] + objs[n - 1].box.max[axis]) * 0.5f;
        while (mid < n - 1 && objs[mid].box.center()[axis] <= mid_world) mid++;
        return mid;
    }
    void build(SortItem *objs, SortItem *tmp, size_t n) {
        int axis = largest_axis(objs, n);
        merge_sort(objs, tmp, n, axis);
        auto branch = [&](SortItem *o, size_t cnt) -> Hittable * {
            owned.emplace_back(new Bvh());
            Bvh *b = owned.back().get();
            b->build(o, tmp, cnt);
            return b;
        };
        if (n == 1) {
            left = objs[0].obj;
            right = nullptr;
        } else if (n == 2) {
            left = objs[0].obj;
            right = objs[1].obj;
        } else if (n == 3) {
            left = branch(objs, 2);
            right = objs[2].obj;
        } else {
            size_t mid = true_middle(objs, n, axis);
            left = branch(objs, mid);
            right = branch(objs + mid, n - mid);
        }
        if (left && right)
            aabb_box = Aabb::surrounding_box(left->bounding_box(), right->bounding_box());
        else
            aabb_box = left->bounding_box();
    }
    bool hit(const Ray &ray, float t_min, float t_max, HitRecord &rec) const override {
        if (tl_detail) tl_detail->node_visits++;
        Vec3 inv_d(recip(ray.dir.x), recip(ray.dir.y), recip(ray.dir.z));
        if (!aabb_box.hit(ray.pos, inv_d, t_min, t_max)) return false;
        HitRecord rr, lr;
        bool rh = right ? right->hit(ray, t_min, t_max, rr) : false;
        bool lh = left ? left->hit(ray, t_min, rh ? rr.t : t_max, lr) : false;
        if (lh) {
            rec = lr;
            return true;
        }
        if (rh) {
            rec = rr;
            return true;
        }
        return false;
    }
    Aabb bounding_box() const override { return aabb_box; }
    void visit_order(std::vector<int32_t> &out) const override {
        if (right) right->visit_order(out);
        if (left) left->visit_order(out);
    }
    size_t count_nodes() const {
        size_t c = 1;
        for (auto &o : owned) c += o->count_nodes();
        return c;
    }
};

Bvh *build_bvh(std::vector<Hittable *> &objs) {
    std::vector<SortItem> items(objs.size()), tmp(objs.size());
    for (size_t i = 0; i < objs.size(); i++) items[i] = SortItem{objs[i], objs[i]->bounding_box()};
    Bvh *b = new Bvh();
    b->build(items.data(), tmp.data(), items.size());
    return b;
}

// src/tracer/mesh.rs
struct Mesh : Hittable {
    std::vector<Triangle> tris;
    std::unique_ptr<Bvh> bvh;
    Mesh(const RtMesh &m, const Material *mat, int32_t first_id) {
        Rotor3 rot = Rotor3::from_euler_angles(m.rotation[2], m.rotation[0], m.rotation[1]).normalized();
        Vec3 scale(m.scale[0], m.scale[1], m.scale[2]), tr(m.translation[0], m.translation[1], m.translation[2]);
        tris.resize(m.n_faces);
        for (uint64_t f = 0; f < m.n_faces; f++) {
            Triangle &t = tris[f];
            for (int k = 0; k < 3; k++) {
                uint64_t vi = m.indices[3 * f + k];
                Vec3 v(m.positions[3 * vi], m.positions[3 * vi + 1], m.positions[3 * vi + 2]);
                t.vertices[k] = rot.rotate(v * scale) + tr;
                Vec3 nn(m.normals[3 * vi], m.normals[3 * vi + 1], m.normals[3 * vi + 2]);
                t.normals[k] = rot.rotate(nn);
            }
            t.material = mat;
            t.two_sided = !m.cull_backface;
            t.prim_id = first_id + (int32_t)f;
        }
        std::vector<Hittable *> objs(tris.size());
        for (size_t i = 0; i < tris.size(); i++) objs[i] = &tris[i];
        bvh.reset(build_bvh(objs));
    }
    bool hit(const Ray &ray, float t_min, float t_max, HitRecord &rec) const override {
        return bvh->hit(ray, t_min, t_max, rec);
    }
    Aabb bounding_box() const override { return bvh->aabb_box; }
    void visit_order(std::vector<int32_t> &out) const override { bvh->visit_order(out); }
};

// src/camera.rs:11-63
struct Camera {
    Vec3 eye, horizontal, vertical, llc, u, v, w;
    float lens_radius;
    Camera() {}
    Camera(const RtCamera &c, float aspect) {
        eye = Vec3(c.eye[0], c.eye[1], c.eye[2]);
        Vec3 lookat(c.lookat[0], c.lookat[1], c.lookat[2]), vup(c.vup[0], c.vup[1], c.vup[2]);
        float h = std::tan((c.fov_degrees * (PI / 180.0f)) / 2.0f);
        float viewport_height = 2.0f * h;
        float viewport_width = aspect * viewport_height;
        w = (eye - lookat).normalized();
        u = vup.cross(w).normalized();
        v = w.cross(u);
        horizontal = (c.focus_dist * viewport_width) * u;
        vertical = (c.focus_dist * viewport_height) * v;
        llc = eye - (horizontal / 2.0f) - (vertical / 2.0f) - c.focus_dist * w;
        lens_radius = c.aperture / 2.0f;
    }
    Ray gen_ray(uint32_t width, uint32_t height, float x, float y, float jx, float jy) const {
        Vec3 rd = lens_radius * in_concentric_disk(jx, jy);
        Vec3 offset = rd.x * u + rd.y * v;
        float s = (x + jy) / (float)(width - 1);
        float t = (y + jx) / (float)(height - 1);
        return Ray(eye + offset, (llc + (s * horizontal) + (t * vertical)) - eye - offset);
    }
};

const float T_MIN = 0.00015f;  // src/render.rs:15
const float T_MAX = 100000.0f; // src/render.rs:16

}  // namespace

struct OracleScene {
    std::vector<Material> materials;
    std::vector<std::unique_ptr<Hittable>> objects;
    std::unique_ptr<Bvh> world;
    RtCamera camera;
    std::vector<float> env;
    uint32_t env_w = 0, env_h = 0;
    int32_t n_prims = 0;
    uint64_t last_rays = 0, last_node_visits = 0, last_prim_tests = 0;
};

namespace {

// src/render.rs:87-104
bool env_pixel(const OracleScene &sc, Vec3 dir, Vec3 *out) {
    if (sc.env.empty()) return false;
    float u = (detm::atan2(dir.x, dir.z) + PI) / (2.0f * PI);
    float v = detm::acos(-dir.y) / PI;
    if (u <= 1.0f && v <= 1.0f) {
        float fx = u * (float)(sc.env_w - 1), fy = (1.0f - v) * (float)(sc.env_h - 1);
        // Rust `as usize`: saturating, NaN -> 0
        uint64_t px = (fx != fx || fx <= 0.0f) ? 0 : (uint64_t)fx;
        uint64_t py = (fy != fy || fy <= 0.0f) ? 0 : (uint64_t)fy;
        const float *p = &sc.env[(py * sc.env_w + px) * 3];
        *out = Vec3(p[0], p[1], p[2]);
        return true;
    }
    return false;
}
// src/render.rs:141-148
Vec3 get_sky(const OracleScene &sc, const Ray &ray, float light_clamp) {
    Vec3 c;
    if (env_pixel(sc, ray.dir, &c)) return c.clamped(splat(0.0f), splat(1.0f) * light_clamp);
    float t = 0.5f * (ray.dir.dot(Vec3(-1.0f, 0.75f, 0.5f).normalized()) + 1.0f);
    return ((1.0f - t) * splat(1.0f) + t * Vec3(0.1f, 0.3f, 0.8f)) * 2.0f;
}

// src/render.rs:58-84
Vec3 ray_color(const OracleScene &sc, Ray ray, uint32_t depth, float light_clamp, uint32_t key) {
    Vec3 color_total = splat(1.0f);
    Ray temp_ray = ray;
    for (uint32_t b = 0; b < depth; b++) {
        HitRecord hit;
        if (tl_counters) tl_counters->rays++;
        if (sc.world->hit(temp_ray, T_MIN, T_MAX, hit)) {
            Scatter s = scatter(*hit.material, temp_ray, hit, key, b);
            if (s.attenuation.component_max() <= 1.0f) {
                color_total = color_total * s.attenuation;
                if (color_total.component_max() < draw(key, b, DIM_RR)) break;
                color_total = color_total * recip(color_total.component_max());
                temp_ray = s.ray;
            } else {
                return color_total * s.attenuation.clamped(splat(0.0f), splat(1.0f) * light_clamp);
            }
        } else {
            return color_total * get_sky(sc, temp_ray, light_clamp);
        }
    }
    return color_total * 0.01f;
}
// src/render.rs:107-120
Vec3 no_bounce(const OracleScene &sc, const Ray &ray, uint32_t key) {
    HitRecord hit;
    if (tl_counters) tl_counters->rays++;
    if (sc.world->hit(ray, T_MIN, T_MAX, hit)) {
        float k = rclamp(Vec3(1.0f, 1.0f, -0.5f).normalized().dot(hit.normal), 0.1f, 1.0f);
        return k * scatter(*hit.material, ray, hit, key, 0).attenuation;
    }
    return get_sky(sc, ray, INFINITY);
}
// src/render.rs:123-129
Vec3 normals_only(const OracleScene &sc, const Ray &ray) {
    HitRecord hit;
    if (tl_counters) tl_counters->rays++;
    if (sc.world->hit(ray, T_MIN, T_MAX, hit)) return (hit.normal + splat(1.0f)) * 0.5f;
    return get_sky(sc, ray, INFINITY);
}
// src/render.rs:132-138
Vec3 colors_only(const OracleScene &sc, const Ray &ray) {
    HitRecord hit;
    if (tl_counters) tl_counters->rays++;
    if (sc.world->hit(ray, T_MIN, T_MAX, hit)) return hit.material->albedo;
    return get_sky(sc, ray, INFINITY);
}

thread_local Counters g_last_counters;

template <class F>
void parallel_chunks(size_t n, int threads, F f) {
    // src/render.rs:17,165: 64 static chunks; here the chunks are handed to `threads` workers
    const size_t chunks = 64;
    size_t chunk = (n + chunks - 1) / chunks;
    if (chunk == 0) chunk = 1;
    std::atomic<size_t> next(0);
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    std::vector<Counters> counters(threads);
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([&, t]() {
            tl_counters = &counters[t];
            tl_detail = g_count_detail.load() ? &counters[t] : nullptr;
            for (;;) {
                size_t c = next.fetch_add(1);
                size_t lo = c * chunk;
                if (lo >= n) break;
                size_t hi = std::min(n, lo + chunk);
                f(lo, hi);
            }
            tl_counters = nullptr;
            tl_detail = nullptr;
        });
    }
    for (auto &th : pool) th.join();
    Counters total;
    for (auto &c : counters) {
        total.rays += c.rays;
        total.node_visits += c.node_visits;
        total.prim_tests += c.prim_tests;
    }
    g_last_counters = total;
}

thread_local std::string g_err;

}  // namespace

extern "C" {

const char *oracle_last_error(void) { return g_err.c_str(); }

// io::load_scene's object loop + Bvh::new: src/io.rs:112-190
int oracle_scene_create(const RtSceneDesc *d, OracleScene **out) {
    if (!d || !out) return RT_ERR_INVALID_ARGUMENT;
    if (d->n_objects == 0) {
        g_err = "empty scene";
        return RT_ERR_EMPTY_SCENE;
    }
    std::unique_ptr<OracleScene> sc(new OracleScene());
    sc->materials.resize(d->n_materials);
    for (uint64_t i = 0; i < d->n_materials; i++) {
        const RtMaterial &m = d->materials[i];
        sc->materials[i] = Material{m.type, Vec3(m.albedo[0], m.albedo[1], m.albedo[2]), m.p0, m.p1};
    }
    int32_t next_id = 0;
    std::vector<Hittable *> world;
    for (uint64_t i = 0; i < d->n_objects; i++) {
        const RtObject &o = d->objects[i];
        if (o.material < 0 || (uint64_t)o.material >= d->n_materials) {
            g_err = "material index out of range";
            return RT_ERR_INVALID_ARGUMENT;
        }
        const Material *mat = &sc->materials[o.material];
        switch (o.type) {
            case RT_OBJ_SPHERE: {
                auto *s = new Sphere();
                s->center = Vec3(o.a[0], o.a[1], o.a[2]);
                s->radius = o.b[0];
                s->material = mat;
                s->prim_id = next_id++;
                sc->objects.emplace_back(s);
                break;
            }
            case RT_OBJ_ABOX:
                sc->objects.emplace_back(new ABox(o.a, o.b, mat, next_id++));
                break;
            case RT_OBJ_CUBE:
                sc->objects.emplace_back(new Cube(o.a, o.b, o.c, mat, next_id++));
                break;
            case RT_OBJ_TRIANGLE: {
                auto *t = new Triangle();
                for (int k = 0; k < 3; k++) {
                    t->vertices[k] = Vec3(o.tri_vertices[3 * k], o.tri_vertices[3 * k + 1], o.tri_vertices[3 * k + 2]);
                    t->normals[k] = Vec3(o.tri_normals[3 * k], o.tri_normals[3 * k + 1], o.tri_normals[3 * k + 2]);
                }
                t->material = mat;
                t->two_sided = o.flag != 0;
                t->prim_id = next_id++;
                sc->objects.emplace_back(t);
                break;
            }
            case RT_OBJ_MESH: {
                if (o.mesh < 0 || (uint64_t)o.mesh >= d->n_meshes) {
                    g_err = "mesh index out of range";
                    return RT_ERR_INVALID_ARGUMENT;
                }
                const RtMesh &m = d->meshes[o.mesh];
                if (m.n_faces == 0) {
                    g_err = "mesh without faces";
                    return RT_ERR_EMPTY_SCENE;
                }
                sc->objects.emplace_back(new Mesh(m, mat, next_id));
                next_id += (int32_t)m.n_faces;
                break;
            }
            default:
                g_err = "unknown object type";
                return RT_ERR_INVALID_ARGUMENT;
        }
        world.push_back(sc->objects.back().get());
    }
    sc->n_prims = next_id;
    sc->world.reset(build_bvh(world));
    sc->camera = d->camera;
    if (d->env.rgb && d->env.width && d->env.height) {
        sc->env.assign(d->env.rgb, d->env.rgb + (size_t)d->env.width * d->env.height * 3);
        sc->env_w = d->env.width;
        sc->env_h = d->env.height;
    }
    *out = sc.release();
    return RT_OK;
}

void oracle_scene_destroy(OracleScene *s) { delete s; }

int64_t oracle_scene_num_prims(const OracleScene *s) { return s->n_prims; }
int64_t oracle_scene_num_nodes(const OracleScene *s) {
    size_t c = s->world->count_nodes();
    for (auto &o : s->objects)
        if (auto *m = dynamic_cast<Mesh *>(o.get())) c += m->bvh->count_nodes();
    return (int64_t)c;
}
// prim ids in the order the reference traversal reaches the leaves
int oracle_visit_order(const OracleScene *s, int32_t *out) {
    std::vector<int32_t> v;
    v.reserve(s->n_prims);
    s->world->visit_order(v);
    std::memcpy(out, v.data(), v.size() * sizeof(int32_t));
    return (int)v.size();
}
void oracle_scene_bounds(const OracleScene *s, float mn[3], float mx[3]) {
    Aabb b = s->world->aabb_box;
    mn[0] = b.min.x, mn[1] = b.min.y, mn[2] = b.min.z;
    mx[0] = b.max.x, mx[1] = b.max.y, mx[2] = b.max.z;
}

// <Bvh as Hittable>::hit on a batch: src/tracer/bvh.rs:143
int oracle_closest_hit_batch(OracleScene *s, const float *org_dir, float t_min, float t_max, size_t n,
                             int32_t *prim_id, float *t, float *normal, float *point, int threads) {
    parallel_chunks(n, threads, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const float *r = org_dir + 6 * i;
            Ray ray(Vec3(r[0], r[1], r[2]), Vec3(r[3], r[4], r[5]));
            HitRecord h;
            if (tl_counters) tl_counters->rays++;
            bool ok = s->world->hit(ray, t_min, t_max, h);
            if (prim_id) prim_id[i] = ok ? h.prim_id : -1;
            if (t) t[i] = ok ? h.t : 0.0f;
            if (normal) {
                normal[3 * i] = ok ? h.normal.x : 0.0f;
                normal[3 * i + 1] = ok ? h.normal.y : 0.0f;
                normal[3 * i + 2] = ok ? h.normal.z : 0.0f;
            }
            if (point) {
                point[3 * i] = ok ? h.point.x : 0.0f;
                point[3 * i + 1] = ok ? h.point.y : 0.0f;
                point[3 * i + 2] = ok ? h.point.z : 0.0f;
            }
        }
    });
    s->last_rays = g_last_counters.rays;
    s->last_node_visits = g_last_counters.node_visits;
    s->last_prim_tests = g_last_counters.prim_tests;
    return RT_OK;
}

// Renderer::render: src/render.rs:162-219
int oracle_render_range(OracleScene *s, const RtRenderParams *p, uint64_t first_pixel, uint64_t n_pixels, float *accum,
                        int threads);
int oracle_render_pass(OracleScene *s, const RtRenderParams *p, float *accum, int threads) {
    if (!p) return RT_ERR_INVALID_ARGUMENT;
    if (p->pixel_count) return oracle_render_range(s, p, p->first_pixel, p->pixel_count, accum, threads);
    return oracle_render_range(s, p, 0, (uint64_t)p->width * p->height, accum, threads);
}
// The same for pixels [first_pixel, first_pixel + n_pixels) of the frame only; `accum` still addresses the
// whole frame.  Lets a test spot-check rows of a frame too large to render on the CPU.
int oracle_render_range(OracleScene *s, const RtRenderParams *p, uint64_t first_pixel, uint64_t n_pixels, float *accum,
                        int threads) {
    if (!s || !p || !accum || p->width < 2 || p->height < 2 || p->samples == 0) return RT_ERR_INVALID_ARGUMENT;
    const uint32_t W = p->width, H = p->height, spp = p->samples;
    if (first_pixel + n_pixels > (uint64_t)W * H) return RT_ERR_INVALID_ARGUMENT;
    Camera cam(s->camera, (float)W / (float)H);
    parallel_chunks((size_t)n_pixels, threads, [&](size_t lo, size_t hi) {
        for (size_t pixel = first_pixel + lo; pixel < first_pixel + hi; pixel++) {
            float x = (float)(pixel % W);
            float y = (float)(H - 1 - pixel / W);
            Vec3 pixel_color(0, 0, 0);
            for (uint32_t k = 0; k < spp; k++) {
                float jx, jy;
                pixel_jitter(p->seed, (uint32_t)pixel, p->pass_index, k, &jx, &jy);
                uint32_t key = path_key(p->seed, (uint32_t)pixel, p->pass_index * spp + k);
                Ray ray = cam.gen_ray(W, H, x, y, jx, jy);
                Vec3 c;
                switch (p->mode) {
                    case RT_MODE_NORMALS: c = normals_only(*s, ray); break;
                    case RT_MODE_COLORS: c = colors_only(*s, ray); break;
                    default: c = ray_color(*s, ray, p->max_bounce, p->light_clamp, key); break;
                }
                pixel_color = pixel_color + c;
                if (!std::isfinite(pixel_color.x)) pixel_color.x = 0.0f;
                if (!std::isfinite(pixel_color.y)) pixel_color.y = 0.0f;
                if (!std::isfinite(pixel_color.z)) pixel_color.z = 0.0f;
            }
            Vec3 add = pixel_color / (float)spp;
            accum[3 * pixel] = accum[3 * pixel] + add.x;
            accum[3 * pixel + 1] = accum[3 * pixel + 1] + add.y;
            accum[3 * pixel + 2] = accum[3 * pixel + 2] + add.z;
        }
    });
    s->last_rays = g_last_counters.rays;
    s->last_node_visits = g_last_counters.node_visits;
    s->last_prim_tests = g_last_counters.prim_tests;
    return RT_OK;
}

// Renderer::preview: src/render.rs:222-264 (one jittered sample per pixel, output overwritten)
int oracle_preview(OracleScene *s, const RtRenderParams *p, float *out, int threads) {
    if (!s || !p || !out || p->width < 2 || p->height < 2) return RT_ERR_INVALID_ARGUMENT;
    const uint32_t W = p->width, H = p->height;
    Camera cam(s->camera, (float)W / (float)H);
    parallel_chunks((size_t)W * H, threads, [&](size_t lo, size_t hi) {
        for (size_t pixel = lo; pixel < hi; pixel++) {
            float x = (float)(pixel % W);
            float y = (float)(H - 1 - pixel / W);
            float jx, jy;
            pixel_jitter(p->seed, (uint32_t)pixel, p->pass_index, 0, &jx, &jy);
            uint32_t key = path_key(p->seed, (uint32_t)pixel, p->pass_index);
            Ray ray = cam.gen_ray(W, H, x, y, jx, jy);
            Vec3 c;
            switch (p->mode) {
                case RT_MODE_NORMALS: c = normals_only(*s, ray); break;
                case RT_MODE_COLORS: c = colors_only(*s, ray); break;
                default: c = no_bounce(*s, ray, key); break;
            }
            out[3 * pixel] = c.x, out[3 * pixel + 1] = c.y, out[3 * pixel + 2] = c.z;
        }
    });
    s->last_rays = g_last_counters.rays;
    return RT_OK;
}

void oracle_set_counting(int enabled) { g_count_detail.store(enabled != 0); }

void oracle_last_counters(const OracleScene *s, uint64_t out[3]) {
    out[0] = s->last_rays, out[1] = s->last_node_visits, out[2] = s->last_prim_tests;
}

// ---- unit-level exports for known-answer tests ----
float oracle_schlick(float cosine, float ni, float nt) { return schlick(cosine, ni, nt); }
void oracle_sky_gradient(const float dir[3], float out[3]) {
    OracleScene sc;
    Ray r;
    r.pos = Vec3(0, 0, 0);
    r.dir = Vec3(dir[0], dir[1], dir[2]);
    Vec3 c = get_sky(sc, r, INFINITY);
    out[0] = c.x, out[1] = c.y, out[2] = c.z;
}
void oracle_sky(const OracleScene *s, const float dir[3], float light_clamp, float out[3]) {
    Ray r;
    r.pos = Vec3(0, 0, 0);
    r.dir = Vec3(dir[0], dir[1], dir[2]);
    Vec3 c = get_sky(*s, r, light_clamp);
    out[0] = c.x, out[1] = c.y, out[2] = c.z;
}
// kind: 0 concentric disk, 1 cosine sphere, 2 uniform sphere, 3 hemisphere about n (r3.. = n)
void oracle_warp(int kind, float r1, float r2, const float n[3], float out[3]) {
    Vec3 v;
    switch (kind) {
        case 0: v = in_concentric_disk(r1, r2); break;
        case 1: v = on_cosine_sphere(r1, r2); break;
        case 2: v = on_uniform_sphere(r1, r2); break;
        default: v = on_hemisphere(Vec3(n[0], n[1], n[2]), r1, r2); break;
    }
    out[0] = v.x, out[1] = v.y, out[2] = v.z;
}
// fn: 0 sin, 1 cos, 2 acos, 3 atan2(x, y), 4 exp
float oracle_math(int fn, float x, float y) {
    switch (fn) {
        case 0: return detm::sin(x);
        case 1: return detm::cos(x);
        case 2: return detm::acos(x);
        case 3: return detm::atan2(x, y);
        default: return detm::exp(x);
    }
}
void oracle_math_array(int fn, const float *x, const float *y, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = oracle_math(fn, x[i], y ? y[i] : 0.0f);
}
float oracle_draw(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t dim) {
    return draw(path_key(seed, pixel, sample), bounce, dim);
}
void oracle_pixel_jitter(uint32_t seed, uint32_t pixel, uint32_t pass, uint32_t s, float out[2]) {
    pixel_jitter(seed, pixel, pass, s, &out[0], &out[1]);
}
// Camera::gen_ray: src/camera.rs:53-63
void oracle_gen_ray(const RtCamera *c, uint32_t W, uint32_t H, float x, float y, float jx, float jy, float out[6]) {
    Camera cam(*c, (float)W / (float)H);
    Ray r = cam.gen_ray(W, H, x, y, jx, jy);
    out[0] = r.pos.x, out[1] = r.pos.y, out[2] = r.pos.z;
    out[3] = r.dir.x, out[4] = r.dir.y, out[5] = r.dir.z;
}
// Material::scatter on one hit: src/material.rs:176
void oracle_scatter(const RtMaterial *m, const float ray_od[6], float t, const float point[3], const float normal[3],
                    uint32_t key, uint32_t bounce, float out_att[3], float out_ray[6]) {
    Material mat{m->type, Vec3(m->albedo[0], m->albedo[1], m->albedo[2]), m->p0, m->p1};
    Ray ray;
    ray.pos = Vec3(ray_od[0], ray_od[1], ray_od[2]);
    ray.dir = Vec3(ray_od[3], ray_od[4], ray_od[5]);
    HitRecord h{t, Vec3(point[0], point[1], point[2]), Vec3(normal[0], normal[1], normal[2]), &mat, 0};
    Scatter s = scatter(mat, ray, h, key, bounce);
    out_att[0] = s.attenuation.x, out_att[1] = s.attenuation.y, out_att[2] = s.attenuation.z;
    out_ray[0] = s.ray.pos.x, out_ray[1] = s.ray.pos.y, out_ray[2] = s.ray.pos.z;
    out_ray[3] = s.ray.dir.x, out_ray[4] = s.ray.dir.y, out_ray[5] = s.ray.dir.z;
}
uint32_t oracle_path_key(uint32_t seed, uint32_t pixel, uint32_t sample) { return path_key(seed, pixel, sample); }

// to_rgb / aces_tonemap: src/render.rs:19-55
uint32_t oracle_to_rgb(const float rgb[3], float gamma) {
    Vec3 c(rgb[0], rgb[1], rgb[2]);
    Vec3 m1c0(0.59719f, 0.07600f, 0.02840f), m1c1(0.35458f, 0.90834f, 0.13383f), m1c2(0.04823f, 0.01566f, 0.83777f);
    Vec3 m2c0(1.60475f, -0.10208f, -0.00327f), m2c1(-0.53108f, 1.10813f, -0.07276f),
        m2c2(-0.07367f, -0.00605f, 1.07602f);
    Vec3 v = m1c0 * c.x + m1c1 * c.y + m1c2 * c.z;
    Vec3 a = v * (v + (splat(1.0f) * 0.0245786f)) - (splat(1.0f) * 0.000090537f);
    Vec3 b = v * (0.983729f * v + (splat(1.0f) * 0.432951f)) + (splat(1.0f) * 0.238081f);
    Vec3 q = a / b;
    Vec3 o = (m2c0 * q.x + m2c1 * q.y + m2c2 * q.z).clamped(splat(0.0f), splat(1.0f));
    // f32::powf; evaluated in double and rounded once, which is what a correctly rounded powf returns except on
    // near-ties of the double result (the product does the same, so the packed pixels can be compared exactly)
    o = Vec3((float)std::pow((double)o.x, (double)gamma), (float)std::pow((double)o.y, (double)gamma),
             (float)std::pow((double)o.z, (double)gamma));
    return (255u << 24) | ((uint32_t)(o.x * 255.4f) << 16) | ((uint32_t)(o.y * 255.4f) << 8) |
           ((uint32_t)(o.z * 255.4f));
}

void oracle_to_rgb_array(const float *rgb, size_t n, float divisor, float gamma, uint32_t *out) {
    for (size_t i = 0; i < n; i++) {
        float c[3] = {rgb[3 * i] / divisor, rgb[3 * i + 1] / divisor, rgb[3 * i + 2] / divisor};
        out[i] = oracle_to_rgb(c, gamma);
    }
}

// bilateral_filter over a whole frame: src/filter.rs:4-48, driven as src/main.rs:117-130.  The index
// arithmetic is the wrapping u32 arithmetic of the reference's release build.
void oracle_bilateral_filter(const float *img, float *out, uint32_t w, uint32_t h, uint32_t diameter, float sigma_i,
                             float sigma_s) {
    // f32::exp as the deterministic polynomial the rest of the restatement uses (the product's kernel uses the same)
    auto gaussian = [](float x, float sigma) { return detm::exp(-(x * x) / (2.0f * sigma * sigma)) / (2.0f * PI * sigma * sigma); };
    const uint32_t half = diameter / 2;
    for (uint32_t cur = 0; cur < w * h; cur++) {
        uint32_t x = cur % w, y = cur / w;
        Vec3 pixel(img[3 * cur], img[3 * cur + 1], img[3 * cur + 2]);
        float current_sum = pixel.x + pixel.y + pixel.z;
        float sum_scale = 1.0f / 3.0f;
        Vec3 filtered(0, 0, 0);
        float pixel_weight = 0.0f;
        for (uint32_t i = 0; i < diameter; i++) {
            for (uint32_t j = 0; j < diameter; j++) {
                uint32_t xn = x - (half - i), yn = y - (half - j);
                xn = xn > w - 1 ? w - 1 : xn;  // clamp(0, w - 1) on a u32
                yn = yn > h - 1 ? h - 1 : yn;
                size_t k = (size_t)yn * w + xn;
                Vec3 neighbor(img[3 * k], img[3 * k + 1], img[3 * k + 2]);
                float neighbor_sum = neighbor.x + neighbor.y + neighbor.z;
                uint32_t dx = x - xn, dy = y - yn;
                float dist = std::sqrt((float)(dx * dx + dy * dy));
                float weight = gaussian(sum_scale * (neighbor_sum - current_sum), sigma_i) * gaussian(dist, sigma_s);
                filtered = filtered + neighbor * weight;
                pixel_weight += weight;
            }
        }
        Vec3 r = filtered / pixel_weight;
        out[3 * cur] = r.x, out[3 * cur + 1] = r.y, out[3 * cur + 2] = r.z;
    }
}

}  // extern "C"
