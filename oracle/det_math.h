/*
 * det_math.h -- TEST INFRASTRUCTURE (part of the CPU oracle; never linked into the product).
 *
 * Single-precision sin/cos/acos/atan2/exp written as fixed sequences of IEEE
 * add/mul/div/sqrt so that every conforming machine returns the same bits.
 * The reference calls the platform libm through Rust's f32::sin, f32::acos, ...
 * (src/random.rs:21,26-29,36-40, src/render.rs:89-90, src/material.rs:107-109);
 * libm results differ between platforms in the last ulp, so the oracle pins the
 * functions instead.  Polynomials and reduction constants are the published
 * Cephes single-precision ones (S. Moshier); accuracy is about 1-2 ulp, checked
 * against libm in tests/test_oracle_math.py.
 *
 * Compile with -ffp-contract=off: no multiply-add here may be fused.
 */
#ifndef ORACLE_DET_MATH_H
#define ORACLE_DET_MATH_H

#include <cmath>
#include <cstdint>
#include <cstring>

namespace detm {

static const float kPi = 3.14159265358979323846f;
static const float kPiO2 = 1.57079632679489661923f;
static const float kPiO4 = 0.78539816339744830962f;

static inline float quiet_nan() { return std::nanf(""); }

/* sin and cos of x (radians), |x| < 1e5; anything else (NaN, inf, huge) gives NaN. */
static inline void sincos(float xx, float *s_out, float *c_out) {
    float x = std::fabs(xx);
    if (!(x < 1.0e5f)) {
        *s_out = quiet_nan();
        *c_out = quiet_nan();
        return;
    }
    int sin_sign = (xx < 0.0f) ? -1 : 1;
    int cos_sign = 1;
    int j = (int)(1.27323954473516f * x); /* 4/pi */
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
    float sv, cv;
    if (j == 1 || j == 2) {
        sv = pc;
        cv = ps;
    } else {
        sv = ps;
        cv = pc;
    }
    *s_out = (sin_sign < 0) ? -sv : sv;
    *c_out = (cos_sign < 0) ? -cv : cv;
}

static inline float sin(float x) {
    float s, c;
    sincos(x, &s, &c);
    return s;
}
static inline float cos(float x) {
    float s, c;
    sincos(x, &s, &c);
    return c;
}

/* asin on |x| <= 0.5 (the only range the callers below need) */
static inline float asin_core(float x) {
    float z = x * x;
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z +
            1.6666752422e-1f) * z * x + x;
}

static inline float acos(float x) {
    if (!(x >= -1.0f && x <= 1.0f)) return quiet_nan();
    if (x < -0.5f) return kPi - 2.0f * asin_core(std::sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin_core(std::sqrt(0.5f * (1.0f - x)));
    return kPiO2 - asin_core(x);
}

static inline float atan(float xx) {
    float x = std::fabs(xx);
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

/* atan2(y, x); NaN in gives NaN out; (0,0) gives 0. */
static inline float atan2(float y, float x) {
    if (x != x || y != y) return quiet_nan();
    if (x > 0.0f) return atan(y / x);
    if (x < 0.0f) return (y >= 0.0f) ? atan(y / x) + kPi : atan(y / x) - kPi;
    if (y > 0.0f) return kPiO2;
    if (y < 0.0f) return -kPiO2;
    return 0.0f;
}

/* exp(x); results below about 1.6e-38 flush to 0. */
static inline float exp(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -87.0f) return 0.0f;
    float z = std::floor(1.44269504088896341f * x + 0.5f);
    x -= z * 0.693359375f;
    x -= z * -2.12194440e-4f;
    int n = (int)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x +
          1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    /* scale by 2^n in two exact steps so n = 128 and n = -126 stay in range */
    int n1 = n / 2, n2 = n - n1;
    uint32_t b1 = (uint32_t)(n1 + 127) << 23, b2 = (uint32_t)(n2 + 127) << 23;
    float f1, f2;
    std::memcpy(&f1, &b1, 4);
    std::memcpy(&f2, &b2, 4);
    return z * f1 * f2;
}

}  // namespace detm
#endif
