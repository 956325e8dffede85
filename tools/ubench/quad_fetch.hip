// quad_fetch.hip -- two ways for every lane of a wave to get one 64-byte tree node of its own.
//
//   mode 0   4 x global_load_dwordx4 per lane, each lane from its own line: 4 line look-ups per node (the closest-hit
//            kernel's node step)
//   mode 1   the four lanes of a quad fetch one node per load instruction -- lane q of the quad reads piece q of the
//            node that lane k owns, in load k -- so a node costs ONE line look-up; afterwards a 4 x 4 transposition
//            inside the quad (two rounds of v_cndmask_b32 with a quad_perm source, 32 instructions, inline assembly) gives every lane
//            the four pieces of its own node.
//   modes 2-6  mode 0's gather with 3 x 16, 2 x 16, 8 x 8, 4 x 12 and 16 x 4 bytes per lane (7 blocks per CU only):
//            what a lane's load costs by width and count
//
// Modes 0 and 1 sum all 16 dwords of every node they fetch; the host checks that the sums agree lane by lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o quad_fetch quad_fetch.hip ; run: ./quad_fetch [table MB] [percent of lanes active]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
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

__host__ __device__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}

// the value of lane (lane ^ 1) / (lane ^ 2) of the same quad
__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); }
template <int K>
__device__ __forceinline__ uint32_t quad_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, false); }

struct Piece {
    uint32_t w[4];
};

__device__ __forceinline__ Piece load_piece(const uint4 *p) {
    const uint4 v = *p;
    return Piece{{v.x, v.y, v.z, v.w}};
}

// Four (x, y) register pairs of two lanes that differ in one bit of their quad place: the lane with the bit clear
// ends up with (x, partner's x), the lane with the bit set with (partner's y, y).  v_cndmask_b32 takes its first
// source through the quad permutation, so an exchange costs one instruction per register.
#define QUAD_SWAP4(PERM, MASK, X0, Y0, X1, Y1, X2, Y2, X3, Y3)                                              \
    {                                                                                                       \
        uint32_t t0, t1, t2, t3;                                                                            \
        asm volatile(                                                                                       \
            "s_nop 1\n\t"                                                                                   \
            "s_mov_b64 vcc, %[m]\n\t"                                                                       \
            "v_cndmask_b32_dpp %[t0], %[x0], %[y0], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"             \
            "v_cndmask_b32_dpp %[t1], %[x1], %[y1], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"             \
            "v_cndmask_b32_dpp %[t2], %[x2], %[y2], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"             \
            "v_cndmask_b32_dpp %[t3], %[x3], %[y3], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"             \
            "s_not_b64 vcc, vcc\n\t"                                                                        \
            "v_cndmask_b32_dpp %[x0], %[y0], %[x0], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"             \
            "v_cndmask_b32_dpp %[x1], %[y1], %[x1], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"             \
            "v_cndmask_b32_dpp %[x2], %[y2], %[x2], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"             \
            "v_cndmask_b32_dpp %[x3], %[y3], %[x3], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"             \
            : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [x0] "+v"(X0), [x1] "+v"(X1), \
              [x2] "+v"(X2), [x3] "+v"(X3)                                                                  \
            : [y0] "v"(Y0), [y1] "v"(Y1), [y2] "v"(Y2), [y3] "v"(Y3), [m] "s"(MASK)                          \
            : "vcc", "scc");                                                                                \
        Y0 = t0, Y1 = t1, Y2 = t2, Y3 = t3;                                                                 \
    }

// r[k] = piece (lane & 3) of the node of quad lane k  ->  r[p] = piece p of this lane's node.  Every lane of the
// quad has to be active.
__device__ __forceinline__ void quad_transpose(Piece r[4]) {
    const unsigned long long bit0 = 0xAAAAAAAAAAAAAAAAull, bit1 = 0xCCCCCCCCCCCCCCCCull;
    QUAD_SWAP4("quad_perm:[1,0,3,2]", bit0, r[0].w[0], r[1].w[0], r[0].w[1], r[1].w[1], r[0].w[2], r[1].w[2], r[0].w[3], r[1].w[3])
    QUAD_SWAP4("quad_perm:[1,0,3,2]", bit0, r[2].w[0], r[3].w[0], r[2].w[1], r[3].w[1], r[2].w[2], r[3].w[2], r[2].w[3], r[3].w[3])
    QUAD_SWAP4("quad_perm:[2,3,0,1]", bit1, r[0].w[0], r[2].w[0], r[0].w[1], r[2].w[1], r[0].w[2], r[2].w[2], r[0].w[3], r[2].w[3])
    QUAD_SWAP4("quad_perm:[2,3,0,1]", bit1, r[1].w[0], r[3].w[0], r[1].w[1], r[3].w[1], r[1].w[2], r[3].w[2], r[1].w[3], r[3].w[3])
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) fetch_kernel(const uint4 *__restrict__ nodes, uint32_t n_nodes, uint32_t active_percent,
                                                       uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < kIters; it++) {
        const uint32_t h = mix(gid * 9781u + it * 0x9E3779B9u);
        const uint32_t node = h & (n_nodes - 1u);  // the table size is a power of two
        const bool active = ((h >> 8) * 0x9E3779B9u >> 16) < active_percent * 655u;
        Piece r[4];
        if (MODE == 0) {
            if (active) {
                const uint4 *p = nodes + (size_t)node * 4;
#pragma unroll
                for (int k = 0; k < 4; k++) r[k] = load_piece(p + k);
            }
        } else if (MODE >= 2) {
            // the same gather with fewer bytes or narrower loads per lane (the sums differ from mode 0's)
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = Piece{{0, 0, 0, 0}};
            if (active) {
                const uint4 *p = nodes + (size_t)node * 4;
                if (MODE == 2) {  // 3 x 16 bytes
#pragma unroll
                    for (int k = 0; k < 3; k++) r[k] = load_piece(p + k);
                } else if (MODE == 3) {  // 2 x 16 bytes
#pragma unroll
                    for (int k = 0; k < 2; k++) r[k] = load_piece(p + k);
                } else if (MODE == 4) {  // 8 x 8 bytes
                    const uint2 *q = (const uint2 *)p;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint2 a = q[2 * k], b = q[2 * k + 1];
                        r[k] = Piece{{a.x, a.y, b.x, b.y}};
                    }
                } else if (MODE == 5) {  // 4 x 12 bytes
                    struct U3 { uint32_t x, y, z; };
                    const U3 *q = (const U3 *)p;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const U3 a = q[k];
                        r[k] = Piece{{a.x, a.y, a.z, 0}};
                    }
                } else {  // 16 x 4 bytes
                    const uint32_t *q = (const uint32_t *)p;
#pragma unroll
                    for (int k = 0; k < 4; k++) r[k] = Piece{{q[4 * k], q[4 * k + 1], q[4 * k + 2], q[4 * k + 3]}};
                }
            }
        } else {
            const uint32_t kNone = 0xffffffffu;
            const uint32_t mine = active ? node * 4u + (lane & 3u) : kNone;   // index of piece 0 of my node, + my place in the quad
            uint32_t owner[4];
            owner[0] = quad_lane<0>(mine), owner[1] = quad_lane<1>(mine), owner[2] = quad_lane<2>(mine), owner[3] = quad_lane<3>(mine);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                r[k] = Piece{{0, 0, 0, 0}};
                // the owner's piece index has the owner's quad place in its low bits: replace them by mine
                if (owner[k] != kNone) r[k] = load_piece(nodes + ((owner[k] & ~3u) | (lane & 3u)));
            }
            quad_transpose(r);
        }
        if (active) {
#pragma unroll
            for (int k = 0; k < 4; k++) acc += r[k].w[0] + r[k].w[1] * 3u + r[k].w[2] * 5u + r[k].w[3] * 7u + (uint32_t)k * r[k].w[0];
        }
    }
    out[gid] = acc;
}

int main(int argc, char **argv) {
    const double table_mb = argc > 1 ? std::atof(argv[1]) : 160.0;
    const uint32_t percent = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 100;
    uint32_t n_nodes = 1;
    while ((double)n_nodes * 2 * 64 <= table_mb * 1024 * 1024) n_nodes *= 2;
    uint4 *nodes;
    CHECK(hipMalloc(&nodes, (size_t)n_nodes * 64));
    {
        std::vector<uint32_t> host((size_t)n_nodes * 16);
        for (size_t i = 0; i < host.size(); i++) host[i] = mix((uint32_t)i + 12345u);
        CHECK(hipMemcpy(nodes, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int blocks_per_cu = 4; blocks_per_cu <= 7; blocks_per_cu += 3) {
        const int blocks = cus * blocks_per_cu;
        const size_t n_threads = (size_t)blocks * kBlock;
        uint32_t *out[2];
        std::vector<uint32_t> res[2];
        for (int mode = 0; mode < 7; mode++) {
            if (mode >= 2) {
                if (blocks_per_cu != 7) break;
                float best = 1e30f;
                for (int rep = 0; rep < 4; rep++) {
                    CHECK(hipEventRecord(e0));
                    if (mode == 2) hipLaunchKernelGGL(fetch_kernel<2>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, percent, out[1]);
                    if (mode == 3) hipLaunchKernelGGL(fetch_kernel<3>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, percent, out[1]);
                    if (mode == 4) hipLaunchKernelGGL(fetch_kernel<4>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, percent, out[1]);
                    if (mode == 5) hipLaunchKernelGGL(fetch_kernel<5>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, percent, out[1]);
                    if (mode == 6) hipLaunchKernelGGL(fetch_kernel<6>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, percent, out[1]);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const double fetched = (double)n_threads * kIters * percent / 100.0;
                static const char *what[] = {"", "", "3 x 16 B", "2 x 16 B", "8 x 8 B", "4 x 12 B", "16 x 4 B"};
                std::printf("mode %d (%s per lane)  %d blocks/CU  %u %% of lanes active  table %g MB: %.3f ms, %.2f clk per node per CU at 2.4 GHz\n",
                            mode, what[mode], blocks_per_cu, percent, table_mb, best, best * 1e-3 * 2.4e9 / (fetched / cus));
                std::printf("{\"mode\": %d, \"loads\": \"%s\", \"blocks_per_cu\": %d, \"active_percent\": %u, \"table_MB\": %g, \"ms\": %.4f, "
                            "\"G_nodes_per_s\": %.2f, \"cus\": %d}\n", mode, what[mode], blocks_per_cu, percent, table_mb, best, fetched / best / 1e6, cus);
                continue;
            }
            CHECK(hipMalloc(&out[mode], n_threads * 4));
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(fetch_kernel<0>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, percent, out[mode]);
                if (mode == 1) hipLaunchKernelGGL(fetch_kernel<1>, dim3(blocks), dim3(kBlock), 0, 0, nodes, n_nodes, percent, out[mode]);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            res[mode].resize(n_threads);
            CHECK(hipMemcpy(res[mode].data(), out[mode], n_threads * 4, hipMemcpyDeviceToHost));
            const double fetched = (double)n_threads * kIters * percent / 100.0;
            std::printf("mode %d  %d blocks/CU  %u %% of lanes active  table %g MB: %.3f ms, %.1f G nodes/s, %.2f clk per node per CU at 2.4 GHz\n",
                        mode, blocks_per_cu, percent, table_mb, best, fetched / best / 1e6, best * 1e-3 * 2.4e9 / (fetched / cus));
            std::printf("{\"mode\": %d, \"loads\": \"%s\", \"blocks_per_cu\": %d, \"active_percent\": %u, \"table_MB\": %g, \"ms\": %.4f, "
                        "\"G_nodes_per_s\": %.2f, \"cus\": %d}\n", mode, mode == 0 ? "4 x 16 B" : "4 x 16 B, one node per quad and load + transposition",
                        blocks_per_cu, percent, table_mb, best, fetched / best / 1e6, cus);
        }
        const bool same = std::memcmp(res[0].data(), res[1].data(), n_threads * 4) == 0;
        std::printf("  sums %s\n", same ? "agree" : "DIFFER");
        CHECK(hipFree(out[0]));
        CHECK(hipFree(out[1]));
        if (!same) return 1;
    }
    return 0;
}
