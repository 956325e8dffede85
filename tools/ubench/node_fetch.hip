// node_fetch.hip -- how fast can the lanes of a wave get one 128-byte tree node each?
//
// The closest-hit kernel fetches a node as seven 16-byte loads per lane, every lane from a line of its own.  This
// microbenchmark times that shape against a cooperative fetch in which eight consecutive lanes read the eight 16-byte
// pieces of ONE node (a coalesced 128-byte request), the pieces land in LDS, and every lane then reads its own node
// from LDS.  Same bytes, same nodes; what differs is the number of (lane, line) requests the vector memory path sees.
//
//   mode 0   7 x global_load_dwordx4 per lane, own line                     (the kernel's node step today)
//   mode 1   cooperative: global_load_lds_dwordx4 (8 lanes per node) + 7 x ds_read_b128 per lane
//   mode 2   cooperative: global_load_dwordx4 into registers + ds_write_b128 + 7 x ds_read_b128 per lane
//
// Build: hipcc --offload-arch=gfx950 -O3 -o node_fetch node_fetch.hip ; run: ./node_fetch [active lanes] [table MB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            std::exit(1);                                                                 \
        }                                                                                 \
    } while (0)

constexpr int kBlock = 256;
constexpr int kIters = 256;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

template <int MODE>
__global__ void __launch_bounds__(kBlock) fetch_kernel(const float4 *__restrict__ nodes, uint32_t n_nodes, uint32_t n_active,
                                                       float *out) {
    // per wave: 64 nodes x 128 B
    __shared__ float4 stage[(MODE == 0 || MODE >= 3) ? 1 : kBlock * 8];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    const bool active = lane < n_active;
    float acc = 0.0f;
    float4 *my_stage = stage + ((MODE == 0 || MODE >= 3) ? 0 : wave * 512);
    for (int it = 0; it < kIters; it++) {
        // modes 3 / 4 / 5: the loads of mode 0 with 8 / 2 / 64 lanes sharing a node (fewer distinct lines per instruction)
        const uint32_t share = MODE == 3 ? (gid >> 3) : (MODE == 4 ? (gid >> 1) : (MODE == 5 ? (gid >> 6) : gid));
        const uint32_t node = mix(share * 9781u + it * 0x9E3779B9u) % n_nodes;
        if (MODE == 0 || MODE >= 3) {
            if (active) {
                const float4 *p = nodes + (size_t)node * 8;
                float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6];
                acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z;
            }
        } else {
            const uint32_t rounds = (n_active + 7u) >> 3;
            for (uint32_t k = 0; k < rounds; k++) {
                const uint32_t q = 8u * k + (lane >> 3);            // which active lane's node this lane helps to fetch
                const uint32_t qn = (uint32_t)__shfl((int)node, (int)q, 64);
                const uint32_t piece = (lane & 7u) ^ (q & 7u);      // the piece that lands at position lane & 7 of the node's slot
                const float4 *src = nodes + (size_t)qn * 8 + piece;
                if (MODE == 1) {
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(my_stage + k * 64), 16, 0, 0);
                } else {
                    const float4 v = *src;
                    my_stage[k * 64 + lane] = v;
                }
            }
            __builtin_amdgcn_s_waitcnt(0);  // everything landed (vmcnt, lgkmcnt)
            __builtin_amdgcn_wave_barrier();
            if (active) {
                const float4 *p = my_stage + lane * 8;
                const uint32_t s = lane & 7u;
                float4 a = p[0 ^ s], b = p[1 ^ s], c = p[2 ^ s], d = p[3 ^ s], e = p[4 ^ s], f = p[5 ^ s], g = p[6 ^ s];
                acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    out[gid] = acc;
}

int main(int argc, char **argv) {
    const uint32_t n_active = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 36u;
    const size_t table_mb = argc > 2 ? (size_t)std::atoi(argv[2]) : 320;
    const uint32_t n_nodes = (uint32_t)(table_mb * 1024 * 1024 / 128);
    float4 *nodes;
    float *out;
    CHECK(hipMalloc(&nodes, (size_t)n_nodes * 128));
    CHECK(hipMemset(nodes, 0, (size_t)n_nodes * 128));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 6; mode++) {
        const bool lds = mode == 1 || mode == 2;
        for (int blocks_per_cu = 3; blocks_per_cu <= 7; blocks_per_cu += (!lds ? 4 : 1)) {
            if (lds && blocks_per_cu > 4) break;  // 32 KB of LDS per block
            const int blocks = cus * blocks_per_cu;
            CHECK(hipMalloc(&out, (size_t)blocks * kBlock * 4));
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(fetch_kernel<0>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, n_active, out);
                if (mode == 1) hipLaunchKernelGGL(fetch_kernel<1>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, n_active, out);
                if (mode == 2) hipLaunchKernelGGL(fetch_kernel<2>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, n_active, out);
                if (mode == 3) hipLaunchKernelGGL(fetch_kernel<3>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, n_active, out);
                if (mode == 4) hipLaunchKernelGGL(fetch_kernel<4>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, n_active, out);
                if (mode == 5) hipLaunchKernelGGL(fetch_kernel<5>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, n_active, out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double fetched = (double)blocks * (kBlock / 64) * n_active * kIters;
            std::printf("mode %d  %d blocks/CU  %u active lanes  table %zu MB: %.3f ms, %.1f G nodes/s, %.2f clk per node per CU at 2.4 GHz\n",
                        mode, blocks_per_cu, n_active, table_mb, best, fetched / best / 1e6,
                        best * 1e-3 * 2.4e9 / (fetched / cus));
            CHECK(hipFree(out));
        }
    }
    return 0;
}
