// rt_device.hpp -- device-side arithmetic of the path-tracing core (gfx950).
//
// Float32 throughout.  The translation unit is compiled with -ffp-contract=off:
// a multiply-add is fused only where it is written as fmaf, which is where the
// reference's vector library (ultraviolet: Vec3::dot / cross / mag_sq) uses
// mul_add.  Division and square root are the correctly rounded ones.  With that,
// every value below is a fixed sequence of IEEE operations, so hit distances,
// normals and whole radiance samples can be compared bit for bit with a CPU
// restatement that performs the same sequence.
#ifndef RT_DEVICE_HPP
#define RT_DEVICE_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_types.h"

namespace rtd {

#define RT_DEV __device__ __forceinline__

constexpr float kPi = 3.14159265358979323846f;
constexpr float kPiO2 = 1.57079632679489661923f;
constexpr float kPiO4 = 0.78539816339744830962f;
constexpr float kAirIndex = 1.00028f;    // src/material.rs:9
constexpr float kBoxTolerance = 0.0001f; // src/tracer/cube.rs:17

struct F3 {
    float x, y, z;
};
RT_DEV F3 f3(float x, float y, float z) { return F3{x, y, z}; }
RT_DEV F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV F3 operator*(F3 a, F3 b) { return F3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV F3 operator*(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
RT_DEV F3 operator*(float s, F3 a) { return F3{s * a.x, s * a.y, s * a.z}; }
RT_DEV F3 operator-(F3 a) { return F3{-a.x, -a.y, -a.z}; }
RT_DEV float dot(F3 a, F3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
RT_DEV F3 cross(F3 a, F3 b) {
    return F3{fmaf(a.y, b.z, -a.z * b.y), fmaf(a.z, b.x, -a.x * b.z), fmaf(a.x, b.y, -a.y * b.x)};
}
RT_DEV float mag_sq(F3 a) { return fmaf(a.x, a.x, fmaf(a.y, a.y, a.z * a.z)); }
RT_DEV F3 normalized(F3 a) {
    float r = 1.0f / sqrtf(mag_sq(a));
    return F3{a.x * r, a.y * r, a.z * r};
}
RT_DEV float max3(F3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
RT_DEV float min3(F3 a) { return fminf(fminf(a.x, a.y), a.z); }
RT_DEV F3 reflected(F3 v, F3 n) { return v - (n * (2.0f * dot(v, n))); }
RT_DEV F3 refracted(F3 v, F3 n, float eta) {
    float ndi = dot(n, v);
    float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
    if (k < 0.0f) return f3(0.0f, 0.0f, 0.0f);
    return v * eta - n * (eta * ndi + sqrtf(k));
}
RT_DEV F3 clamped(F3 v, float lo, float hi) {
    return F3{fminf(fmaxf(v.x, lo), hi), fminf(fmaxf(v.y, lo), hi), fminf(fmaxf(v.z, lo), hi)};
}
// Rust's f32::clamp: NaN stays NaN
RT_DEV float rclamp(float x, float lo, float hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
RT_DEV float powi5(float x) {
    float x2 = x * x;
    return (x2 * x2) * x;
}

// ---- rotor sandwich (ultraviolet Rotor3::rotate_vec) ----
struct Rotor {
    float s, xy, xz, yz;
};
RT_DEV F3 rotate(Rotor r, F3 v) {
    float fx = r.s * v.x + r.xy * v.y + r.xz * v.z;
    float fy = r.s * v.y - r.xy * v.x + r.yz * v.z;
    float fz = r.s * v.z - r.xz * v.x - r.yz * v.y;
    float fw = r.xy * v.z - r.xz * v.y + r.yz * v.x;
    return F3{r.s * fx + r.xy * fy + r.xz * fz + r.yz * fw, r.s * fy - r.xy * fx - r.xz * fw + r.yz * fz,
              r.s * fz + r.xy * fw - r.xz * fx - r.yz * fy};
}
RT_DEV Rotor reversed(Rotor r) { return Rotor{r.s, -r.xy, -r.xz, -r.yz}; }

// ---- elementary functions as fixed operation sequences (Cephes single-precision
// polynomials); arguments outside the stated range give NaN ----
RT_DEV float qnan() { return __int_as_float(0x7fc00000); }

RT_DEV void det_sincos(float xx, float *s_out, float *c_out) {
    float x = fabsf(xx);
    if (!(x < 1.0e5f)) {
        *s_out = qnan();
        *c_out = qnan();
        return;
    }
    int sin_sign = (xx < 0.0f) ? -1 : 1;
    int cos_sign = 1;
    int j = (int)(1.27323954473516f * x);
    if (j & 1) j += 1;
    float y = (float)j;
    j &= 7;
    if (j > 3) {
        sin_sign = -sin_sign;
        cos_sign = -cos_sign;
        j -= 4;
    }
    if (j > 1) cos_sign = -cos_sign;
    x = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = x * x;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z -
               0.5f * z + 1.0f;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
    bool swap = (j == 1 || j == 2);
    float sv = swap ? pc : ps;
    float cv = swap ? ps : pc;
    *s_out = (sin_sign < 0) ? -sv : sv;
    *c_out = (cos_sign < 0) ? -cv : cv;
}
RT_DEV float det_asin_core(float x) {
    float z = x * x;
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z +
            1.6666752422e-1f) * z * x + x;
}
RT_DEV float det_acos(float x) {
    if (!(x >= -1.0f && x <= 1.0f)) return qnan();
    if (x < -0.5f) return kPi - 2.0f * det_asin_core(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * det_asin_core(sqrtf(0.5f * (1.0f - x)));
    return kPiO2 - det_asin_core(x);
}
RT_DEV float det_atan(float xx) {
    float x = fabsf(xx);
    float y;
    if (x > 2.414213562373095f) {
        y = kPiO2;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = kPiO4;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return (xx < 0.0f) ? -y : y;
}
RT_DEV float det_atan2(float y, float x) {
    if (x != x || y != y) return qnan();
    if (x > 0.0f) return det_atan(y / x);
    if (x < 0.0f) return (y >= 0.0f) ? det_atan(y / x) + kPi : det_atan(y / x) - kPi;
    if (y > 0.0f) return kPiO2;
    if (y < 0.0f) return -kPiO2;
    return 0.0f;
}
RT_DEV float det_exp(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return __int_as_float(0x7f800000);
    if (x < -87.0f) return 0.0f;
    float z = floorf(1.44269504088896341f * x + 0.5f);
    x -= z * 0.693359375f;
    x -= z * -2.12194440e-4f;
    int n = (int)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x +
          1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    int n1 = n / 2, n2 = n - n1;
    float f1 = __int_as_float((n1 + 127) << 23), f2 = __int_as_float((n2 + 127) << 23);
    return z * f1 * f2;
}

// ---- counter-based random numbers: a pure function of (seed, pixel, sample, bounce, dimension) ----
RT_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
RT_DEV uint32_t path_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t k = mix32(seed ^ 0xA511E9B3U);
    k = mix32(k ^ pixel);
    k = mix32(k + sample * 0x9E3779B9U);
    return k;
}
RT_DEV float u01(uint32_t bits) { return (float)(bits >> 9) * (1.0f / 8388608.0f); }
RT_DEV float draw(uint32_t key, uint32_t bounce, uint32_t dim) {
    uint32_t ctr = bounce * 8U + dim;
    return u01(mix32(key ^ mix32(ctr * 0x9E3779B9U + 0x7F4A7C15U)));
}
enum { DIM_R1 = 0, DIM_R2 = 1, DIM_S0 = 2, DIM_S1 = 3, DIM_S2 = 4, DIM_CHOICE = 5, DIM_RR = 6 };
// R2 (plastic-number) additive recurrence in 32-bit fixed point, start hashed per (pixel, pass)
RT_DEV void pixel_jitter(uint32_t seed, uint32_t pixel, uint32_t pass, uint32_t s, float *jx, float *jy) {
    uint32_t pk = mix32(mix32(mix32(seed ^ 0x243F6A88U) ^ pixel) + pass * 0x9E3779B9U);
    uint32_t sx = mix32(pk + 0x68E31DA4U), sy = mix32(pk + 0xB5297A4DU);
    uint32_t x = sx + (s + 1U) * 0xC13FA9A9U;
    uint32_t y = sy + (s + 1U) * 0x91E10DA5U;
    *jx = u01(x);
    *jy = u01(y);
}

// ---- sampling warps (src/random.rs) ----
RT_DEV F3 in_concentric_disk(float r1, float r2) {  // :14-22
    float a = (2.0f * r1) - 1.0f, b = (2.0f * r2) - 1.0f;
    float radius, phi;
    if ((a * a) > (b * b)) {
        radius = a;
        phi = (kPi / 4.0f) * (b / a);
    } else {
        radius = b;
        phi = (kPi / 2.0f) - ((kPi / 4.0f) * (a / b));
    }
    float sn, cs;
    det_sincos(phi, &sn, &cs);
    return f3(cs * radius, sn * radius, 0.0f);
}
RT_DEV F3 on_cosine_sphere(float r1, float r2) {  // :25-32
    float z = sqrtf(1.0f - r2);
    float phi = 2.0f * kPi * r1;
    float sn, cs;
    det_sincos(phi, &sn, &cs);
    float x = cs * sqrtf(r2);
    float y = sn * sqrtf(r2);
    return f3(x, y, z);
}
RT_DEV F3 on_uniform_sphere(float r1, float r2) {  // :35-43
    float phi = 2.0f * kPi * r1;
    float theta = det_acos((2.0f * r2) - 1.0f);
    float sp, cp, st, ct;
    det_sincos(phi, &sp, &cp);
    det_sincos(theta, &st, &ct);
    return f3(cp * st, sp * st, ct);
}
RT_DEV F3 in_unit_sphere(float r1, float r2, float u) { return on_uniform_sphere(r1, r2) * u; }  // :8-11
// Onb::from_w + local (src/ray.rs:35-50), quasirandom_on_hemisphere (src/random.rs:46-48)
RT_DEV F3 on_hemisphere(F3 normal, float r1, float r2) {
    F3 w = normalized(normal);
    F3 a = (fabsf(w.x) > 0.9f) ? f3(0.0f, 1.0f, 0.0f) : f3(1.0f, 0.0f, 0.0f);
    F3 v = normalized(cross(w, a));
    F3 u = cross(w, v);
    F3 l = on_cosine_sphere(r1, r2);
    return l.x * u + l.y * v + l.z * w;
}

// schlick (src/material.rs:90-94)
RT_DEV float schlick(float cosine, float ni, float nt) {
    float q = (ni - nt) / (ni + nt);
    float r0 = q * q;
    return rclamp(r0 + (1.0f - r0) * powi5(1.0f - cosine), 0.0f, 1.0f);
}

struct Ray {
    F3 o, d;
};
RT_DEV F3 ray_at(const Ray &r, float t) { return r.o + t * r.d; }

// Material::scatter (src/material.rs:176-183 -> :28-35, :44-59, :67-81, :96-136).
// Returns the attenuation; new_dir is the scattered direction before Ray::new normalises it.
//
// The lanes of a wave hold different materials, and a wave runs every branch one of its lanes takes, so
// the work the models have in common is written once: the cosine-weighted direction (Lambertian,
// Glossy), the normal jittered by a point of the unit ball (Glossy, Metal, Dielectric; the three draws
// differ per model), the mirror direction and the Fresnel term.  Each lane still evaluates exactly its
// model's expressions on its model's arguments.
RT_DEV F3 scatter(const RtDevMaterial &m, const Ray &ray, float t, F3 normal, uint32_t key, uint32_t bounce,
                  F3 *new_dir) {
    const float r1 = draw(key, bounce, DIM_R1), r2 = draw(key, bounce, DIM_R2);
    const F3 albedo = f3(m.albedo[0], m.albedo[1], m.albedo[2]);
    const bool lambertian = m.type == RT_MAT_LAMBERTIAN, glossy = m.type == RT_MAT_GLOSSY;
    const bool metal = m.type == RT_MAT_METAL, dielectric = !(lambertian || glossy || metal);
    F3 diffuse_dir = f3(0.0f, 0.0f, 0.0f);
    if (lambertian || glossy) diffuse_dir = on_hemisphere(normal, r1, r2);
    if (lambertian) {
        *new_dir = diffuse_dir;
        return albedo;
    }
    // Glossy / Metal: in_unit_sphere() draws three numbers (src/material.rs:46,69); Dielectric: the
    // caller's r1, r2 and one more (src/material.rs:98-99)
    const float s0 = draw(key, bounce, DIM_S0), s1 = draw(key, bounce, DIM_S1), s2 = draw(key, bounce, DIM_S2);
    const F3 rnd = in_unit_sphere(dielectric ? r1 : s0, dielectric ? r2 : s1, dielectric ? s0 : s2);
    const float roughness = metal ? m.p0 : m.p1;
    const F3 jn = normalized(normal + (rnd * roughness));
    F3 outward = jn, color = f3(1.0f, 1.0f, 1.0f) * 0.9f;
    float ni = kAirIndex, nt, cosine;
    bool total_reflection = false;
    if (dielectric) {
        const float ior = m.p0;
        if (dot(ray.d, normal) > 0.0f) {
            F3 absorbance = albedo * -t * 2.0f;
            outward = -jn;
            ni = ior;
            nt = kAirIndex;
            cosine = dot(ray.d, jn);
            color = f3(det_exp(absorbance.x), det_exp(absorbance.y), det_exp(absorbance.z));
        } else {
            nt = ior;
            cosine = dot(-ray.d, jn);
        }
        total_reflection = !((ni / nt) * sqrtf(1.0f - (cosine * cosine)) <= 1.0f);
    } else {
        nt = 1.0f + m.p0;  // Glossy: 1 + reflectance; unused by Metal
        cosine = dot(-ray.d, jn);
    }
    const F3 mirrored = reflected(ray.d, outward);
    if (metal) {
        *new_dir = mirrored;
        return clamped(albedo + (f3(1.0f, 1.0f, 1.0f) - albedo) * powi5(1.0f - cosine), 0.0f, 1.0f);
    }
    const float prob = schlick(cosine, ni, nt);
    const bool specular = total_reflection || draw(key, bounce, DIM_CHOICE) <= prob;
    if (glossy) {
        *new_dir = specular ? mirrored : diffuse_dir;
        return specular ? color : albedo;
    }
    *new_dir = specular ? mirrored : refracted(ray.d, outward, ni / nt);
    return color;
}

// get_sky / get_pixel_from_vec (src/render.rs:87-104, :141-148)
RT_DEV F3 get_sky(const RtSceneView &sc, F3 dir, float light_clamp) {
    if (sc.env) {
        float u = (det_atan2(dir.x, dir.z) + kPi) / (2.0f * kPi);
        float v = det_acos(-dir.y) / kPi;
        if (u <= 1.0f && v <= 1.0f) {
            float fx = u * (float)(sc.env_w - 1), fy = (1.0f - v) * (float)(sc.env_h - 1);
            // `as usize`: saturating, NaN -> 0
            uint32_t px = (fx != fx || fx <= 0.0f) ? 0u : (uint32_t)fx;
            uint32_t py = (fy != fy || fy <= 0.0f) ? 0u : (uint32_t)fy;
            const float *p = sc.env + ((size_t)py * sc.env_w + px) * 3;
            return clamped(f3(p[0], p[1], p[2]), 0.0f, 1.0f * light_clamp);
        }
    }
    float t = 0.5f * (dot(dir, normalized(f3(-1.0f, 0.75f, 0.5f))) + 1.0f);
    return ((1.0f - t) * f3(1.0f, 1.0f, 1.0f) + t * f3(0.1f, 0.3f, 0.8f)) * 2.0f;
}

// Camera::gen_ray (src/camera.rs:53-63); the returned direction is not yet normalised
RT_DEV void gen_ray(const RtDevCamera &c, uint32_t width, uint32_t height, float x, float y, float jx, float jy,
                    F3 *org, F3 *dir) {
    F3 u = f3(c.u[0], c.u[1], c.u[2]), v = f3(c.v[0], c.v[1], c.v[2]);
    F3 eye = f3(c.eye[0], c.eye[1], c.eye[2]);
    F3 hor = f3(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
    F3 ver = f3(c.vertical[0], c.vertical[1], c.vertical[2]);
    F3 llc = f3(c.llc[0], c.llc[1], c.llc[2]);
    F3 rd = c.lens_radius * in_concentric_disk(jx, jy);
    F3 offset = rd.x * u + rd.y * v;
    float s = (x + jy) / (float)(width - 1);
    float t = (y + jx) / (float)(height - 1);
    *org = eye + offset;
    *dir = (llc + (s * hor) + (t * ver)) - eye - offset;
}

}  // namespace rtd
#endif
