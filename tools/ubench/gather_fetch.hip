// Calibration of rocprofv3's FETCH_SIZE on the access patterns of the closest-hit kernel.
// Three kernels read a known number of bytes from a table far larger than the 256 MiB Infinity Cache:
//   gather128  every lane reads one whole 128-byte record (7 x 16 B used + 16 B pad, as a tree node) at a random index
//   gather48   every lane reads one 48-byte record (3 x 16 B, as a primitive) at a random index
//   stream16   lane i reads 16 B at base + 16 i: the coalesced stream FETCH_SIZE is documented for
// Run under `rocprofv3 --pmc FETCH_SIZE --output-format csv`; tools/fetch_calibration.py divides the bytes
// printed here by the counter.  Build: hipcc --offload-arch=gfx950 -O3 -o gather_fetch gather_fetch.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
            return 1;                                                                \
        }                                                                            \
    } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) fill(float4 *t, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) t[i] = make_float4((float)(i & 1023), 1.0f, 2.0f, 3.0f);
}

// one 128-byte record per lane and turn; seven of its eight 16-byte words are read, as the node step does
__global__ void __launch_bounds__(256) gather128(const float4 *__restrict__ t, uint32_t n_records, uint32_t turns, float *out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    for (uint32_t k = 0; k < turns; k++) {
        const uint32_t r = mix32(gid * 0x9E3779B9U + k * 0x85EBCA6BU + 1u) % n_records;
        const float4 *p = t + (size_t)r * 8;
        float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6];
        acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z;
    }
    if (acc == -1.0f) out[gid] = acc;  // never true: keeps the loads alive
}

__global__ void __launch_bounds__(256) gather48(const float4 *__restrict__ t, uint32_t n_records, uint32_t turns, float *out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    for (uint32_t k = 0; k < turns; k++) {
        const uint32_t r = mix32(gid * 0x9E3779B9U + k * 0x85EBCA6BU + 7u) % n_records;
        const float4 *p = t + (size_t)r * 3;
        float4 a = p[0], b = p[1], c = p[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == -1.0f) out[gid] = acc;
}

__global__ void __launch_bounds__(256) stream16(const float4 *__restrict__ t, size_t n, float *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.0f;
    for (; i < n; i += stride) {
        float4 a = t[i];
        acc += a.x + a.y + a.z + a.w;
    }
    if (acc == -1.0f) out[0] = acc;
}

int main(int argc, char **argv) {
    const size_t table_bytes = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 4096ull) << 20;  // MiB
    const uint32_t turns = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 16;
    float4 *table = nullptr;
    float *out = nullptr;
    CHECK(hipMalloc((void **)&table, table_bytes));
    const uint32_t lanes = 256u * 256u * 16u;  // 1 M lanes
    CHECK(hipMalloc((void **)&out, lanes * sizeof(float)));
    const size_t n16 = table_bytes / 16;
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, table, n16);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms = 0.0f;
    const uint32_t n128 = (uint32_t)(table_bytes / 128), n48 = (uint32_t)(table_bytes / 48);
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather128, dim3(lanes / 256), dim3(256), 0, 0, table, n128, turns, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("{\"kernel\": \"gather128\", \"records\": %llu, \"bytes_used\": %llu, \"bytes_lines\": %llu, \"ms\": %.3f}\n",
                    (unsigned long long)lanes * turns, (unsigned long long)lanes * turns * 112ull,
                    (unsigned long long)lanes * turns * 128ull, ms);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather48, dim3(lanes / 256), dim3(256), 0, 0, table, n48, turns, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // a 48-byte record at a multiple of 48 bytes straddles two 128-byte lines for 2 of every 8 records (offsets 96 and 112): 1.25 lines each
        std::printf("{\"kernel\": \"gather48\", \"records\": %llu, \"bytes_used\": %llu, \"bytes_lines\": %llu, \"ms\": %.3f}\n",
                    (unsigned long long)lanes * turns, (unsigned long long)lanes * turns * 48ull,
                    (unsigned long long)lanes * turns * 128ull * 10ull / 8ull, ms);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream16, dim3(2048), dim3(256), 0, 0, table, n16, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("{\"kernel\": \"stream16\", \"records\": %llu, \"bytes_used\": %llu, \"bytes_lines\": %llu, \"ms\": %.3f}\n",
                    (unsigned long long)n16, (unsigned long long)table_bytes, (unsigned long long)table_bytes, ms);
    }
    CHECK(hipFree(table));
    CHECK(hipFree(out));
    return 0;
}
