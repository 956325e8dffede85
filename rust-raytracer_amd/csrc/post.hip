// post.hip -- display-side kernels that follow the path-tracing passes, so that a progressive
// display never reads float buffers back: ACES tonemap + gamma + pack (aces_tonemap / to_rgb,
// src/render.rs:39-55, applied to buffer / pass as src/main.rs:266-271 does) and one iteration of
// the bilateral filter (src/filter.rs:14-48).  Both are plain streaming kernels: one lane per
// pixel, HBM-bound (12 B read + 4 B written per pixel for the tonemap).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>

#include "../../include/rtcore.h"
#include "rt_device.hpp"

namespace {

constexpr int kBlock = 256;

__global__ void __launch_bounds__(kBlock)
    tonemap_kernel(const float *__restrict__ rgb, size_t n, float divisor, float inv_gamma, uint32_t *__restrict__ argb) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // *color / pass (Vec3 / f32 divides each component)
    float cx = rgb[3 * i] / divisor, cy = rgb[3 * i + 1] / divisor, cz = rgb[3 * i + 2] / divisor;
    // M1 * color: columns (0.59719, 0.07600, 0.02840), (0.35458, 0.90834, 0.13383), (0.04823, 0.01566, 0.83777)
    float v[3] = {0.59719f * cx + 0.35458f * cy + 0.04823f * cz, 0.07600f * cx + 0.90834f * cy + 0.01566f * cz,
                  0.02840f * cx + 0.13383f * cy + 0.83777f * cz};
    float q[3];
    for (int k = 0; k < 3; k++) {
        float a = v[k] * (v[k] + 0.0245786f) - 0.000090537f;
        float b = v[k] * (0.983729f * v[k] + 0.432951f) + 0.238081f;
        q[k] = a / b;
    }
    // M2 * (a / b): columns (1.60475, -0.10208, -0.00327), (-0.53108, 1.10813, -0.07276), (-0.07367, -0.00605, 1.07602)
    float o[3] = {1.60475f * q[0] + -0.53108f * q[1] + -0.07367f * q[2], -0.10208f * q[0] + 1.10813f * q[1] + -0.00605f * q[2],
                  -0.00327f * q[0] + -0.07276f * q[1] + 1.07602f * q[2]};
    uint32_t out = 255u << 24;
    for (int k = 0; k < 3; k++) {
        float c = fminf(fmaxf(o[k], 0.0f), 1.0f);  // clamped: NaN -> 0
        // f32::powf of the reference, taken through double precision: the device's and the host's single-precision
        // pow differ in the last bit now and then, their double-precision ones round to the same float
        c = (float)pow((double)c, (double)inv_gamma);
        float s = c * 255.4f;
        uint32_t u = (s != s || s <= 0.0f) ? 0u : (uint32_t)s;  // `as u32` saturates, NaN -> 0
        out |= u << (16 - 8 * k);
    }
    argb[i] = out;
}

__device__ __forceinline__ float gaussian(float x, float sigma) {
    const float kPi = 3.14159265358979323846f;
    // the same exp polynomial as the path tracer's (and the oracle's), so the filtered frame can be compared exactly
    return rtd::det_exp(-(x * x) / (2.0f * sigma * sigma)) / (2.0f * kPi * sigma * sigma);
}

__global__ void __launch_bounds__(kBlock)
    bilateral_kernel(const float *__restrict__ img, float *__restrict__ out, uint32_t w, uint32_t h, uint32_t diameter,
                     float sigma_i, float sigma_s) {
    uint32_t cur = blockIdx.x * blockDim.x + threadIdx.x;
    if (cur >= w * h) return;
    uint32_t x = cur % w, y = cur / w;
    const float *p = img + (size_t)cur * 3;
    float current_sum = p[0] + p[1] + p[2];
    float f0 = 0.0f, f1 = 0.0f, f2 = 0.0f, weight_sum = 0.0f;
    const uint32_t half = diameter / 2;
    for (uint32_t i = 0; i < diameter; i++) {
        for (uint32_t j = 0; j < diameter; j++) {
            // wrapping u32 arithmetic as the reference's release build: a negative x - half + i wraps to a
            // huge value and clamps to the far edge (src/filter.rs:33-34)
            uint32_t xn = x - (half - i), yn = y - (half - j);
            if (xn > w - 1) xn = w - 1;
            if (yn > h - 1) yn = h - 1;
            const float *q = img + ((size_t)yn * w + xn) * 3;
            float neighbor_sum = q[0] + q[1] + q[2];
            uint32_t dx = x - xn, dy = y - yn;
            float dist = sqrtf((float)(dx * dx + dy * dy));
            float wgt = gaussian((1.0f / 3.0f) * (neighbor_sum - current_sum), sigma_i) * gaussian(dist, sigma_s);
            f0 += q[0] * wgt, f1 += q[1] * wgt, f2 += q[2] * wgt;
            weight_sum += wgt;
        }
    }
    out[(size_t)cur * 3] = f0 / weight_sum;
    out[(size_t)cur * 3 + 1] = f1 / weight_sum;
    out[(size_t)cur * 3 + 2] = f2 / weight_sum;
}

}  // namespace

extern "C" int rt_set_error_message(int code, const char *msg);  // rtcore.hip

extern "C" {

int rt_tonemap_device(const float *d_rgb, size_t n_pixels, float divisor, float inv_gamma, uint32_t *d_argb,
                      void *stream) {
    if (!d_rgb || !d_argb) return rt_set_error_message(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (n_pixels == 0) return RT_OK;
    if (n_pixels > 0x7fffffffull) return rt_set_error_message(RT_ERR_INVALID_ARGUMENT, "frame too large");
    hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((n_pixels + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, d_rgb, n_pixels, divisor, inv_gamma, d_argb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rt_set_error_message(RT_ERR_DEVICE, hipGetErrorString(e));
    return RT_OK;
}

int rt_bilateral_filter_device(const float *d_in, float *d_out, uint32_t width, uint32_t height, uint32_t diameter,
                               float sigma_i, float sigma_s, void *stream) {
    if (!d_in || !d_out || d_in == d_out) return rt_set_error_message(RT_ERR_INVALID_ARGUMENT, "null or aliased buffers");
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7fffffffull)
        return rt_set_error_message(RT_ERR_INVALID_ARGUMENT, "bad frame size");
    hipLaunchKernelGGL(bilateral_kernel, dim3((width * height + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, d_in, d_out, width, height, diameter, sigma_i, sigma_s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rt_set_error_message(RT_ERR_DEVICE, hipGetErrorString(e));
    return RT_OK;
}

// Host-buffer variants: copy in, run the kernels above on the current device, copy out, synchronise.
// The staging buffers are kept per device and only ever grow, so a host that tonemaps every pass
// (src/main.rs:266-271) pays for the allocations once; one call at a time (the mutex is held across the call).
namespace {
struct Staging {
    void *in = nullptr, *out = nullptr;
    size_t in_bytes = 0, out_bytes = 0;
};
std::mutex g_staging_mutex;
Staging g_staging[64];

int grow(void **p, size_t *have, size_t want) {
    if (want <= *have) return RT_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {
        *p = nullptr;
        return rt_set_error_message(e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_DEVICE, hipGetErrorString(e));
    }
    *have = want;
    return RT_OK;
}
int staging_for(size_t in_bytes, size_t out_bytes, Staging **st) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return rt_set_error_message(RT_ERR_DEVICE, hipGetErrorString(e));
    if (dev < 0 || dev >= 64) return rt_set_error_message(RT_ERR_DEVICE, "device index out of range");
    Staging *s = &g_staging[dev];
    int rc = grow(&s->in, &s->in_bytes, in_bytes);
    if (rc == RT_OK) rc = grow(&s->out, &s->out_bytes, out_bytes);
    *st = s;
    return rc;
}
}  // namespace

int rt_tonemap(const float *rgb, size_t n_pixels, float divisor, float inv_gamma, uint32_t *argb) {
    if (!rgb || !argb) return rt_set_error_message(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (n_pixels == 0) return RT_OK;
    std::lock_guard<std::mutex> lock(g_staging_mutex);
    Staging *st = nullptr;
    int rc = staging_for(n_pixels * 12, n_pixels * 4, &st);
    hipError_t e;
    if (rc == RT_OK && (e = hipMemcpy(st->in, rgb, n_pixels * 12, hipMemcpyHostToDevice)) != hipSuccess)
        rc = rt_set_error_message(RT_ERR_DEVICE, hipGetErrorString(e));
    if (rc == RT_OK) rc = rt_tonemap_device((const float *)st->in, n_pixels, divisor, inv_gamma, (uint32_t *)st->out, nullptr);
    if (rc == RT_OK && (e = hipMemcpy(argb, st->out, n_pixels * 4, hipMemcpyDeviceToHost)) != hipSuccess)
        rc = rt_set_error_message(RT_ERR_DEVICE, hipGetErrorString(e));
    return rc;
}

int rt_bilateral_filter(const float *in, float *out, uint32_t width, uint32_t height, uint32_t diameter, float sigma_i,
                        float sigma_s) {
    if (!in || !out || in == out) return rt_set_error_message(RT_ERR_INVALID_ARGUMENT, "null or aliased buffers");
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7fffffffull)
        return rt_set_error_message(RT_ERR_INVALID_ARGUMENT, "bad frame size");
    const size_t bytes = (size_t)width * height * 12;
    std::lock_guard<std::mutex> lock(g_staging_mutex);
    Staging *st = nullptr;
    int rc = staging_for(bytes, bytes, &st);
    hipError_t e;
    if (rc == RT_OK && (e = hipMemcpy(st->in, in, bytes, hipMemcpyHostToDevice)) != hipSuccess)
        rc = rt_set_error_message(RT_ERR_DEVICE, hipGetErrorString(e));
    if (rc == RT_OK)
        rc = rt_bilateral_filter_device((const float *)st->in, (float *)st->out, width, height, diameter, sigma_i, sigma_s, nullptr);
    if (rc == RT_OK && (e = hipMemcpy(out, st->out, bytes, hipMemcpyDeviceToHost)) != hipSuccess)
        rc = rt_set_error_message(RT_ERR_DEVICE, hipGetErrorString(e));
    return rc;
}

}  // extern "C"
