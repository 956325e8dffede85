// compact.hip -- makes the compact nodes and the leaf blocks (compact.hpp, rt_types.h) from a built tree.
//
// Per node and axis the grid starts at the lowest plane of the node's children and its step is the smallest power of
// two with 255 steps reaching the highest; a child's low planes are rounded down to the grid and its high planes up,
// so a compact box contains the exact one.  An axis on which some child's box is not finite (a NaN or infinite vertex)
// or which no such grid spans gets an infinite step: the traversal then works that axis out plane by plane and the
// not-a-number planes are ignored, as the reference's box test ignores them (src/tracer/cube.rs:163-171).
//
// A leaf (one or two primitives; blocks repeat for more) becomes one 128-byte block: the exact box that guards the
// leaf slot, the visit rank of its first primitive, and the primitive records, so that a leaf step reads one line.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>

#include <rocprim/device/device_scan.hpp>

#include "../../include/rtcore.h"
#include "compact.hpp"
#include "rt_types.h"

namespace rtcore {
namespace {

constexpr int kBlock = 256;

#define CT_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) {                                                   \
            err = std::string(#call) + ": " + hipGetErrorString(e_);             \
            return RT_ERR_DEVICE;                                                 \
        }                                                                         \
    } while (0)

// blocks_of[first primitive of a leaf] = number of blocks the leaf takes (two primitives per block)
__global__ void __launch_bounds__(kBlock) leaf_sizes_kernel(const float4 *__restrict__ nodes, uint32_t n_nodes, uint32_t *blocks_of) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes) return;
    const float4 cwf = nodes[(size_t)i * RT_NODE_FLOAT4S + 6];
    const uint32_t cw[4] = {__float_as_uint(cwf.x), __float_as_uint(cwf.y), __float_as_uint(cwf.z), __float_as_uint(cwf.w)};
    for (int c = 0; c < 4; c++)
        if (cw[c] != RT_CHILD_NONE && (cw[c] & RT_CHILD_LEAF))
            blocks_of[cw[c] & RT_LEAF_FIRST_MASK] = (((cw[c] >> 28) & 7u) + 2u) >> 1;
}

__device__ __forceinline__ float lane_of(const float4 v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

__global__ void __launch_bounds__(kBlock)
    compact_kernel(const float4 *__restrict__ nodes, uint32_t n_nodes, const float4 *__restrict__ prims,
                   const uint32_t *__restrict__ block_at, float4 *__restrict__ qnodes, float4 *__restrict__ leaves) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes) return;
    const float4 *nd = nodes + (size_t)i * RT_NODE_FLOAT4S;
    float4 plane[6];
    for (int k = 0; k < 6; k++) plane[k] = nd[k];
    const float4 cwf = nd[6];
    uint32_t cw[4] = {__float_as_uint(cwf.x), __float_as_uint(cwf.y), __float_as_uint(cwf.z), __float_as_uint(cwf.w)};
    float origin[3], step[3];
    uint32_t qlo[3], qhi[3];
    for (int a = 0; a < 3; a++) {
        double mn = 1e300, mx = -1e300;
        bool open = false, any = false;
        for (int c = 0; c < 4; c++) {
            if (cw[c] == RT_CHILD_NONE) continue;
            any = true;
            const float lo = lane_of(plane[a], c), hi = lane_of(plane[a + 3], c);
            if (!(fabsf(lo) <= 3.0e38f) || !(fabsf(hi) <= 3.0e38f) || !(lo <= hi))
                open = true;
            else
                mn = fmin(mn, (double)lo), mx = fmax(mx, (double)hi);
        }
        origin[a] = 0.0f, step[a] = __int_as_float(0x7f800000), qlo[a] = 0u, qhi[a] = 0u;
        if (!any || open) continue;
        const double ext = mx - mn;
        double s = 0x1p-100;
        if (ext > 0.0) {
            int e;
            (void)frexp(ext / 255.0, &e);  // ext / 255 = f * 2^e with f in [0.5, 1)
            s = ldexp(1.0, e - 1);
            if (s < 0x1p-100) s = 0x1p-100;
            while (s * 255.0 < ext) s *= 2.0;
        }
        if (s > 0x1p100) continue;
        origin[a] = (float)mn;
        step[a] = (float)s;
        for (int c = 0; c < 4; c++) {
            if (cw[c] == RT_CHILD_NONE) continue;
            double l = floor(((double)lane_of(plane[a], c) - mn) / s), h = ceil(((double)lane_of(plane[a + 3], c) - mn) / s);
            l = fmin(fmax(l, 0.0), 255.0), h = fmin(fmax(h, 0.0), 255.0);
            qlo[a] |= (uint32_t)l << (8 * c);
            qhi[a] |= (uint32_t)h << (8 * c);
        }
    }
    // the leaf blocks, and the child words that name them
    for (int c = 0; c < 4; c++) {
        if (cw[c] == RT_CHILD_NONE || !(cw[c] & RT_CHILD_LEAF)) continue;
        const uint32_t first = cw[c] & RT_LEAF_FIRST_MASK, count = ((cw[c] >> 28) & 7u) + 1u;
        const uint32_t block = block_at[first];
        const float4 g0 = make_float4(lane_of(plane[0], c), lane_of(plane[1], c), lane_of(plane[2], c), lane_of(plane[3], c));
        for (uint32_t k = 0; k < count; k += 2) {
            float4 *b = leaves + (size_t)(block + (k >> 1)) * RT_LEAF_BLOCK_FLOAT4S;
            const uint32_t here = count - k < 2u ? 1u : 2u;
            b[0] = g0;
            b[1] = make_float4(lane_of(plane[4], c), lane_of(plane[5], c), __uint_as_float(first + k), __uint_as_float(here));
            const float4 *r = prims + (size_t)(first + k) * RT_PRIM_FLOAT4S;
            const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            b[2] = r[0], b[3] = r[1], b[4] = r[2];
            b[5] = here > 1u ? r[3] : zero, b[6] = here > 1u ? r[4] : zero, b[7] = here > 1u ? r[5] : zero;
        }
        cw[c] = RT_CHILD_LEAF | ((count - 1u) << 28) | block;
    }
    float4 *q = qnodes + (size_t)i * RT_QNODE_FLOAT4S;
    q[0] = make_float4(origin[0], origin[1], origin[2], step[0]);
    q[1] = make_float4(step[1], step[2], __uint_as_float(qlo[0]), __uint_as_float(qlo[1]));
    q[2] = make_float4(__uint_as_float(qlo[2]), __uint_as_float(qhi[0]), __uint_as_float(qhi[1]), __uint_as_float(qhi[2]));
    q[3] = make_float4(__uint_as_float(cw[0]), __uint_as_float(cw[1]), __uint_as_float(cw[2]), __uint_as_float(cw[3]));
}

struct Scratch {  // freed on every way out
    void *p = nullptr;
    ~Scratch() {
        if (p) (void)hipFree(p);
    }
};

}  // namespace

void CompactTree::release() {
    if (qnodes) (void)hipFree(qnodes);
    if (leaves) (void)hipFree(leaves);
    qnodes = leaves = nullptr;
    qnodes_bytes = leaves_bytes = 0;
    n_blocks = 0;
}

int compact_tree(const void *nodes, uint64_t n_nodes, const void *prims, uint64_t n_prims, CompactTree &out, std::string &err) {
    out = CompactTree();
    if (!nodes || !prims || n_nodes == 0 || n_prims == 0 || n_nodes > 0x7fffffffull || n_prims > RT_LEAF_FIRST_MASK) {
        err = "compact_tree: nothing to do or too large";
        return RT_ERR_INVALID_ARGUMENT;
    }
    const uint32_t nn = (uint32_t)n_nodes, np = (uint32_t)n_prims;
    const uint32_t node_grid = (nn + kBlock - 1) / kBlock;
    // leaf sizes by first primitive, their running sum = each leaf's first block
    Scratch sizes, at, tmp;
    const size_t count_bytes = ((size_t)np + 1) * sizeof(uint32_t);
    CT_HIP(hipMalloc(&sizes.p, count_bytes));
    CT_HIP(hipMalloc(&at.p, count_bytes));
    CT_HIP(hipMemsetAsync(sizes.p, 0, count_bytes, 0));
    hipLaunchKernelGGL(leaf_sizes_kernel, dim3(node_grid), dim3(kBlock), 0, 0, (const float4 *)nodes, nn, (uint32_t *)sizes.p);
    CT_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    CT_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, (uint32_t *)sizes.p, (uint32_t *)at.p, 0u, (size_t)np + 1, rocprim::plus<uint32_t>(), 0));
    CT_HIP(hipMalloc(&tmp.p, tmp_bytes ? tmp_bytes : 16));
    CT_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, (uint32_t *)sizes.p, (uint32_t *)at.p, 0u, (size_t)np + 1, rocprim::plus<uint32_t>(), 0));
    uint32_t n_blocks = 0;
    CT_HIP(hipMemcpy(&n_blocks, (const uint32_t *)at.p + np, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (n_blocks == 0 || n_blocks > RT_LEAF_FIRST_MASK) {
        err = "compact_tree: the leaf blocks outgrow a child word";
        return RT_ERR_INVALID_ARGUMENT;
    }
    CompactTree t;
    t.qnodes_bytes = (size_t)nn * (RT_QNODE_FLOAT4S * 16) + 128;
    t.leaves_bytes = (size_t)n_blocks * (RT_LEAF_BLOCK_FLOAT4S * 16) + 128;
    t.n_blocks = n_blocks;
    hipError_t e = hipMalloc(&t.qnodes, t.qnodes_bytes);
    if (e == hipSuccess) e = hipMalloc(&t.leaves, t.leaves_bytes);
    if (e != hipSuccess) {
        t.release();
        err = "compact_tree: hipMalloc of " + std::to_string(t.qnodes_bytes + t.leaves_bytes) + " bytes failed";
        return RT_ERR_OUT_OF_MEMORY;
    }
    hipLaunchKernelGGL(compact_kernel, dim3(node_grid), dim3(kBlock), 0, 0, (const float4 *)nodes, nn, (const float4 *)prims,
                       (const uint32_t *)at.p, (float4 *)t.qnodes, (float4 *)t.leaves);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        t.release();
        err = std::string("compact_kernel: ") + hipGetErrorString(e);
        return RT_ERR_DEVICE;
    }
    out = t;
    return RT_OK;
}

}  // namespace rtcore
