// gpu_build.hip -- the scene build on the device (opt-in: RT_BUILD_DEVICE / RTCORE_BUILD=gpu).
//
// The default build restates Bvh::new (src/tracer/bvh.rs:39-117) on the host, because the reference's tree --
// and with it the winner among hits at exactly the same distance -- is defined by a sequential merge sort under a
// comparator that is not a strict weak order.  A parallel build cannot reproduce that permutation.  What it can
// keep is everything else the traversal's answer depends on: the primitive records are the same bits (same
// operations as the host build, -ffp-contract=off), every leaf is guarded by a box that contains its primitives,
// and interior boxes contain their children.  So this builder gives the same (primitive, t) as the exact build for
// every ray whose closest hit is not an exact tie between two primitives, up to the rays that graze a guard box
// within rounding (the reference's guard boxes are the EPSILON-padded unions of Aabb::surrounding_box,
// src/tracer/cube.rs:179-193; here a two-primitive leaf gets the same padded union and a single primitive its own
// box, as in the reference's one-child nodes, src/tracer/bvh.rs:46-49).
//
// Pipeline (one stream; the host reads a counter or two per tree level and does nothing else between the stages):
//   1. boxes      primitive boxes in declaration order (mesh faces transformed as Mesh::new does,
//                 src/tracer/mesh.rs:31-73), the box of every scene object, scene bounds
//   2. splits     the binary tree, level by level, by the reference's own rule (median of the longest axis of the
//                 union box; scene objects are split as wholes until a segment holds one mesh; a segment whose centres
//                 all lie on one side gives up its extreme item, as Bvh::true_middle does).  See td_* below.
//                 RTCORE_DEVICE_TREE=morton selects the older form instead: 63-bit Morton codes, radix sort, Karras'
//                 binary radix tree -- 25 % more node visits per ray, kept for comparison.
//   3. emit       records, shade records and leaf boxes in array order (array index = "visit rank" of this tree)
//   4. fit        boxes bottom-up, one launch per level; the same pass fills the table that picks the roots of the
//                 4-wide nodes: the cheapest summed half area of wide-node root boxes for handing a subtree to at
//                 most j slots (the cost model of the host build's collapse)
//   5. collapse   top-down, one launch per level of the wide tree; wide nodes are numbered level by level, within a
//                 level in the order of their parents (block-wide prefix sum, one atomic per block)
#include <hip/hip_runtime.h>

#include <string.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "gpu_build.hpp"
#include "rt_device.hpp"
#include "scene_build.hpp"

namespace rtcore {
namespace {

using rtd::F3;
using rtd::f3;
using rtd::Rotor;

constexpr int kBlock = 256;
constexpr float kEps = 1.1920929e-7f;  // f32::EPSILON
constexpr uint32_t kLeafBit = 0x80000000u;

struct MeshDev {       // one mesh OBJECT (a mesh array may serve several objects)
    float rot[4];      // rotor s, xy, xz, yz
    float scale[3];
    uint32_t meta;     // primitive meta word of its faces
    float tr[3];
    uint32_t n_vertices;
    uint64_t vtx_off;  // first vertex in the vertex pools
    uint64_t face_off; // first face in the index pool
};
struct SimpleDev {  // a sphere, box, rotated box or lone triangle: record and box made on the host
    uint32_t rec[12];
    float nrm[12];
    float box[8];  // lo.xyz, pad, hi.xyz, pad
};
struct Sources {
    const uint32_t *seg_first;  // [n_seg] first primitive id of each object, ascending
    const uint32_t *seg_ref;    // [n_seg] bit 31: mesh, low bits: index into meshes / simples
    uint32_t n_seg;
    const MeshDev *meshes;
    const SimpleDev *simples;
    const float *pos, *nrm;  // vertex pools (xyz triples)
    const uint32_t *idx;     // index pool (three per face)
};

__device__ __forceinline__ float fmin_(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
__device__ __forceinline__ float fmax_(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }

__device__ __forceinline__ uint32_t find_segment(const Sources &S, uint32_t p) {
    uint32_t lo = 0, hi = S.n_seg;  // largest s with seg_first[s] <= p
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (S.seg_first[mid] <= p)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// The three transformed corners and normals of face f of a mesh object (src/tracer/mesh.rs:49-73).
__device__ __forceinline__ void mesh_face(const Sources &S, const MeshDev &m, uint32_t f, F3 v[3], F3 n[3], uint32_t *bad) {
    const Rotor rot{m.rot[0], m.rot[1], m.rot[2], m.rot[3]};
    const F3 sc = f3(m.scale[0], m.scale[1], m.scale[2]), tr = f3(m.tr[0], m.tr[1], m.tr[2]);
    const uint32_t *ix = S.idx + 3 * (m.face_off + f);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t vi = ix[k];
        if (vi >= m.n_vertices) {
            *bad = 1;
            vi = 0;
        }
        const float *P = S.pos + 3 * (m.vtx_off + vi), *N = S.nrm + 3 * (m.vtx_off + vi);
        v[k] = rtd::rotate(rot, f3(P[0], P[1], P[2]) * sc) + tr;
        n[k] = rtd::rotate(rot, f3(N[0], N[1], N[2]));
    }
}

__device__ __forceinline__ uint32_t ordered(float f) {  // float order -> unsigned order
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__host__ __device__ inline float unordered(uint32_t u) {
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// ---- 1. boxes + scene bounds ----
// bounds[0..2] = min of lo (ordered-integer form), bounds[3..5] = max of hi; status[0] = 1 on a bad vertex index
// obj_of[p] = the scene object primitive p belongs to; obj_bounds[6 * object ..] = that object's box (ordered-integer form)
__global__ void __launch_bounds__(kBlock) boxes_kernel(Sources S, uint32_t n, float4 *box_lo, float4 *box_hi, uint32_t *bounds,
                                                       uint32_t *status, uint32_t *obj_of, uint32_t *obj_bounds) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    const float inf = __int_as_float(0x7f800000);
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    uint32_t s = 0xFFFFFFFFu;
    if (p < n) {
        s = find_segment(S, p);
        obj_of[p] = s;
        const uint32_t ref = S.seg_ref[s];
        if (ref & kLeafBit) {
            const MeshDev &m = S.meshes[ref & ~kLeafBit];
            F3 v[3], nn[3];
            uint32_t bad = 0;
            mesh_face(S, m, p - S.seg_first[s], v, nn, &bad);
            if (bad) status[0] = 1;
            // Triangle::bounding_box (src/tracer/triangle.rs:80-105)
            lo[0] = fmin_(v[0].x, fmin_(v[1].x, v[2].x)), hi[0] = fmax_(v[0].x, fmax_(v[1].x, v[2].x));
            lo[1] = fmin_(v[0].y, fmin_(v[1].y, v[2].y)), hi[1] = fmax_(v[0].y, fmax_(v[1].y, v[2].y));
            lo[2] = fmin_(v[0].z, fmin_(v[1].z, v[2].z)), hi[2] = fmax_(v[0].z, fmax_(v[1].z, v[2].z));
        } else {
            const SimpleDev &o = S.simples[ref];
            for (int k = 0; k < 3; k++) lo[k] = o.box[k], hi[k] = o.box[4 + k];
        }
        box_lo[p] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        box_hi[p] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    // block reduce (NaN-ignoring; an infinite or NaN box does not move the bounds)
    __shared__ float red[6][kBlock / 64];
    {  // the object's box: one set of atomics per wave while the wave lies in one object
        const unsigned long long m = __ballot(p < n);
        if (m != 0ull) {
            const uint32_t first = (uint32_t)__shfl((int)s, __ffsll((long long)m) - 1, 64);
            const bool uniform = __all(p >= n || s == first);
            for (int k = 0; k < 3; k++) {
                float a = (fabsf(lo[k]) < inf) ? lo[k] : inf, b = (fabsf(hi[k]) < inf) ? hi[k] : -inf;
                if (uniform) {
                    for (int off = 32; off > 0; off >>= 1) a = fminf(a, __shfl_xor(a, off, 64)), b = fmaxf(b, __shfl_xor(b, off, 64));
                    if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m) - 1))
                        atomicMin(&obj_bounds[6 * first + k], ordered(a)), atomicMax(&obj_bounds[6 * first + 3 + k], ordered(b));
                } else if (p < n) {
                    atomicMin(&obj_bounds[6 * s + k], ordered(a)), atomicMax(&obj_bounds[6 * s + 3 + k], ordered(b));
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float a = (fabsf(lo[k]) < inf) ? lo[k] : inf, b = (fabsf(hi[k]) < inf) ? hi[k] : -inf;
        for (int off = 32; off > 0; off >>= 1) {
            a = fminf(a, __shfl_down(a, off, 64));
            b = fmaxf(b, __shfl_down(b, off, 64));
        }
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = a, red[3 + k][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = red[k][0];
        for (int w = 1; w < kBlock / 64; w++) v = (k < 3) ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
        if (k < 3)
            atomicMin(&bounds[k], ordered(v));
        else
            atomicMax(&bounds[k], ordered(v));
    }
}

// ---- 2. Morton codes ----
__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // bit i -> bit 3i
    uint64_t x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__global__ void __launch_bounds__(kBlock) morton_kernel(const float4 *box_lo, const float4 *box_hi, uint32_t n, const uint32_t *bounds,
                                                        uint64_t *keys, uint32_t *vals) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const float4 lo = box_lo[p], hi = box_hi[p];
    const float c[3] = {0.5f * lo.x + 0.5f * hi.x, 0.5f * lo.y + 0.5f * hi.y, 0.5f * lo.z + 0.5f * hi.z};
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float b0 = unordered(bounds[k]), b1 = unordered(bounds[3 + k]);
        const float ext = b1 - b0;
        float u = (ext > 0.0f) ? (c[k] - b0) / ext : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);  // (NaN -> 0)
        q[k] = (uint32_t)fminf(u * 2097152.0f, 2097151.0f);
    }
    // Descending order in space and, among equal codes (the radix sort is stable), in declaration order: the reference
    // visits the child with the larger coordinates first (right, then left), and its stable sort leaves the earlier
    // declared of two coincident primitives on the left, where it is visited last and wins an exact tie
    // (src/tracer/bvh.rs:42,148-155).  Array index = visit rank, so ranks here grow the same way.
    const uint64_t code = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    keys[n - 1 - p] = ~code & 0x7FFFFFFFFFFFFFFFull;
    vals[n - 1 - p] = p;
}

// ---- 3. records in sorted order ----
__global__ void __launch_bounds__(kBlock) emit_kernel(Sources S, uint32_t n, const uint32_t *perm, const float4 *box_lo,
                                                      const float4 *box_hi, float4 *prims, float4 *shade, float4 *leaf_lo,
                                                      float4 *leaf_hi) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = perm[i];
    const uint32_t s = find_segment(S, p);
    const uint32_t ref = S.seg_ref[s];
    float4 r0, r1, r2, n0, n1, n2;
    if (ref & kLeafBit) {
        const MeshDev &m = S.meshes[ref & ~kLeafBit];
        F3 v[3], nn[3];
        uint32_t bad = 0;
        mesh_face(S, m, p - S.seg_first[s], v, nn, &bad);
        const F3 e1 = v[1] - v[0], e2 = v[2] - v[0];
        r0 = make_float4(v[0].x, v[0].y, v[0].z, e1.x);
        r1 = make_float4(e1.y, e1.z, e2.x, e2.y);
        r2 = make_float4(e2.z, 0.0f, __uint_as_float(m.meta), __uint_as_float(p));
        n0 = make_float4(nn[0].x, nn[0].y, nn[0].z, 0.0f);
        n1 = make_float4(nn[1].x, nn[1].y, nn[1].z, 0.0f);
        n2 = make_float4(nn[2].x, nn[2].y, nn[2].z, 0.0f);
    } else {
        const float4 *q = (const float4 *)&S.simples[ref];
        r0 = q[0], r1 = q[1], r2 = q[2], n0 = q[3], n1 = q[4], n2 = q[5];
    }
    prims[3 * (size_t)i + 0] = r0, prims[3 * (size_t)i + 1] = r1, prims[3 * (size_t)i + 2] = r2;
    float4 *sh = shade + 6 * (size_t)i;
    sh[0] = r0, sh[1] = r1, sh[2] = r2, sh[3] = n0, sh[4] = n1, sh[5] = n2;
    leaf_lo[i] = box_lo[p];
    leaf_hi[i] = box_hi[p];
}

// ---- 4. binary radix tree (Karras 2012) over sorted keys; equal keys are told apart by their position ----
__device__ __forceinline__ int prefix_len(const uint64_t *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)(i ^ j));
    return __clzll((long long)(a ^ b));
}
// child / parent words: bit 31 set = leaf i, clear = internal node i
__global__ void __launch_bounds__(kBlock) hierarchy_kernel(const uint64_t *keys, int n, uint2 *children, uint32_t *parent_of_node,
                                                           uint32_t *parent_of_leaf) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (prefix_len(keys, n, i, i + 1) - prefix_len(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = prefix_len(keys, n, i, i - d);
    int lmax = 2;
    while (prefix_len(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (prefix_len(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = prefix_len(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (prefix_len(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    const uint32_t left = (first == gamma) ? (kLeafBit | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t right = (last == gamma + 1) ? (kLeafBit | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    children[i] = make_uint2(left, right);
    if (left & kLeafBit) parent_of_leaf[gamma] = (uint32_t)i; else parent_of_node[gamma] = (uint32_t)i;
    if (right & kLeafBit) parent_of_leaf[gamma + 1] = (uint32_t)i; else parent_of_node[gamma + 1] = (uint32_t)i;
}

// ---- 4b. the reference's own way of splitting, level by level (the default tree of this build) ----
// Bvh::new splits a set of primitives along the axis on which the union of their boxes is longest, at the middle of
// that extent: the primitives whose centre lies beyond the middle go to one child, the others to the other
// (src/tracer/bvh.rs:83-117; its sort decides the order inside each side, which only matters for exact ties).  The same
// rule here, all segments of a level at once: union boxes by atomics (one set per wave while a wave lies in one
// segment), one flag per primitive, one prefix sum over the flags, one stable scatter.  The side with the larger
// coordinates goes first in the array, so array order is the reference's visit order (right child first), and the
// array starts in reverse declaration order, so the earlier declared of two coincident primitives ends up later, as
// under the reference's stable sort.  A segment whose primitives all fall on one side is cut in the middle of its
// range instead.  Segments of one primitive are leaves; every other segment becomes a binary node.
constexpr uint32_t kNoSegment = 0xFFFFFFFFu;
struct TdNode {
    uint32_t begin, end;      // positions [begin, end) of the primitive array
    uint32_t lo[3], hi[3];    // union box of the primitives' boxes, ordered-integer form
    uint32_t n_front;         // primitives that go to the front child
    uint32_t cut_in_middle;   // all on one side: cut by position
    uint32_t front, back;     // segment (binary node) of each child, kNoSegment for a single primitive
    uint32_t obj_lo, obj_hi;  // smallest and largest scene object among its primitives
    uint32_t one_sided;       // 1: every centre lies at or below the middle, 2: every centre lies beyond it, 0: neither
    uint32_t peels;           // splits in a row above this segment that split off one item
    unsigned long long key_max, key_min;  // (ordered sort key << 32 | object or primitive) of the extreme items, for one-sided segments
};

// Axis and middle of a segment as Bvh::largest_axis / true_middle take them (src/tracer/bvh.rs:97-117).
__device__ __forceinline__ void td_axis_mid(const TdNode &nd, int *axis, float *mid) {
    const float l[3] = {unordered(nd.lo[0]), unordered(nd.lo[1]), unordered(nd.lo[2])};
    const float h[3] = {unordered(nd.hi[0]), unordered(nd.hi[1]), unordered(nd.hi[2])};
    const float w = h[0] - l[0], ht = h[1] - l[1], d = h[2] - l[2];
    *axis = (w > ht && w > d) ? 0 : (ht > d ? 1 : 2);
    *mid = (l[*axis] + h[*axis]) * 0.5f;
}
// The box a primitive answers with on `axis`: its object's while the segment holds several objects, else its own.
__device__ __forceinline__ void td_item_box(const TdNode &nd, uint32_t p, int axis, const float4 *box_lo, const float4 *box_hi,
                                            const uint32_t *obj_of, const uint32_t *obj_bounds, float *lo, float *hi, uint32_t *who) {
    if (nd.obj_lo != nd.obj_hi) {
        const uint32_t o = obj_of[p];
        *lo = unordered(obj_bounds[6 * o + axis]), *hi = unordered(obj_bounds[6 * o + 3 + axis]), *who = o;
    } else {
        const float4 bl = box_lo[p], bh = box_hi[p];
        *lo = axis == 0 ? bl.x : (axis == 1 ? bl.y : bl.z), *hi = axis == 0 ? bh.x : (axis == 1 ? bh.y : bh.z), *who = p;
    }
}

__global__ void __launch_bounds__(kBlock) td_init_kernel(uint32_t n, uint32_t *idx, uint32_t *seg_of, TdNode *nodes) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) idx[i] = n - 1u - i, seg_of[i] = 0u;
    if (i == 0) {
        TdNode r;
        r.begin = 0, r.end = n, r.n_front = 0, r.cut_in_middle = 0, r.front = r.back = kNoSegment;
        for (int k = 0; k < 3; k++) r.lo[k] = 0xFFFFFFFFu, r.hi[k] = 0u;
        r.obj_lo = 0xFFFFFFFFu, r.obj_hi = 0u, r.one_sided = 0, r.peels = 0, r.key_max = 0ull, r.key_min = ~0ull;
        nodes[0] = r;
    }
}

// One item of the union: this lane's box and object, then the segmented reduction over the wave and the atomics.
struct TdAcc {
    float lo[3], hi[3];
    uint32_t o_lo, o_hi;
};
__device__ __forceinline__ void td_acc_clear(TdAcc &a) {
    const float inf = __int_as_float(0x7f800000);
    for (int k = 0; k < 3; k++) a.lo[k] = inf, a.hi[k] = -inf;
    a.o_lo = 0xFFFFFFFFu, a.o_hi = 0u;
}
__device__ __forceinline__ void td_acc_add(TdAcc &a, uint32_t p, const float4 *box_lo, const float4 *box_hi, const uint32_t *obj_of) {
    const float inf = __int_as_float(0x7f800000);
    const float4 l = box_lo[p], h = box_hi[p];
    const float lo[3] = {l.x, l.y, l.z}, hi[3] = {h.x, h.y, h.z};
    for (int k = 0; k < 3; k++) {  // a NaN or infinite box does not move the union
        if (fabsf(lo[k]) < inf) a.lo[k] = fminf(a.lo[k], lo[k]);
        if (fabsf(hi[k]) < inf) a.hi[k] = fmaxf(a.hi[k], hi[k]);
    }
    const uint32_t o = obj_of[p];
    a.o_lo = min(a.o_lo, o), a.o_hi = max(a.o_hi, o);
}
__device__ __forceinline__ void td_acc_atomics(const TdAcc &a, TdNode &nd) {
    for (int k = 0; k < 3; k++) atomicMin(&nd.lo[k], ordered(a.lo[k])), atomicMax(&nd.hi[k], ordered(a.hi[k]));
    atomicMin(&nd.obj_lo, a.o_lo), atomicMax(&nd.obj_hi, a.o_hi);
}
// Positions are ordered by segment, so the lanes of one segment are consecutive: a segmented minimum / maximum over the
// wave (the windows of the doubling steps may overlap, which these operations do not mind), then one set of atomics per
// run of lanes instead of one per lane.  Every lane of the wave must call it; seg = kNoSegment for a lane with nothing.
__device__ __forceinline__ void td_acc_flush_wave(TdAcc a, uint32_t seg, TdNode *nodes) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t other_seg = (uint32_t)__shfl_down((int)seg, off, 64);
        const bool take = lane + off < 64u && other_seg == seg;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float x = __shfl_down(a.lo[k], off, 64), y = __shfl_down(a.hi[k], off, 64);
            if (take) a.lo[k] = fminf(a.lo[k], x), a.hi[k] = fmaxf(a.hi[k], y);
        }
        const uint32_t x = (uint32_t)__shfl_down((int)a.o_lo, off, 64), y = (uint32_t)__shfl_down((int)a.o_hi, off, 64);
        if (take) a.o_lo = min(a.o_lo, x), a.o_hi = max(a.o_hi, y);
    }
    const uint32_t before = (uint32_t)__shfl_up((int)seg, 1, 64);
    if (seg != kNoSegment && (lane == 0u || before != seg)) td_acc_atomics(a, nodes[seg]);  // the first lane of a run holds its result
}

__global__ void __launch_bounds__(kBlock) td_bounds_kernel(uint32_t n, const uint32_t *idx, const uint32_t *seg_of, const float4 *box_lo,
                                                           const float4 *box_hi, const uint32_t *obj_of, TdNode *nodes) {
    const uint32_t pos = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t seg = pos < n ? seg_of[pos] : kNoSegment;
    if (__ballot(seg != kNoSegment) == 0ull) return;
    TdAcc a;
    td_acc_clear(a);
    if (seg != kNoSegment) td_acc_add(a, idx[pos], box_lo, box_hi, obj_of);
    td_acc_flush_wave(a, seg, nodes);
}
// The same for the levels near the root, where a few segments hold millions of primitives each and one set of atomics
// per wave would queue up on a handful of addresses: a thread walks kTdCoarse positions a block apart and keeps its
// union while the segment stays the same.
constexpr uint32_t kTdCoarse = 32;
__global__ void __launch_bounds__(kBlock) td_bounds_coarse_kernel(uint32_t n, const uint32_t *idx, const uint32_t *seg_of,
                                                                  const float4 *box_lo, const float4 *box_hi, const uint32_t *obj_of,
                                                                  TdNode *nodes) {
    const uint32_t base = blockIdx.x * (kBlock * kTdCoarse) + threadIdx.x;
    TdAcc a;
    td_acc_clear(a);
    uint32_t cur = kNoSegment;
    for (uint32_t it = 0; it < kTdCoarse; it++) {
        const uint32_t pos = base + it * kBlock;
        const uint32_t seg = pos < n ? seg_of[pos] : kNoSegment;
        if (seg == kNoSegment) continue;
        if (seg != cur) {
            if (cur != kNoSegment) td_acc_atomics(a, nodes[cur]);  // (rare near the root)
            td_acc_clear(a);
            cur = seg;
        }
        td_acc_add(a, idx[pos], box_lo, box_hi, obj_of);
    }
    td_acc_flush_wave(a, cur, nodes);
}

// While a segment holds primitives of several scene objects the objects are what is split (the world's tree of
// src/io.rs:188-190 is built over whole objects, a mesh being one of them): every primitive answers with its object's box.
__global__ void __launch_bounds__(kBlock) td_classify_kernel(uint32_t n, const uint32_t *idx, const uint32_t *seg_of, const float4 *box_lo,
                                                             const float4 *box_hi, const uint32_t *obj_of, const uint32_t *obj_bounds,
                                                             const TdNode *nodes, uint32_t *flags) {
    const uint32_t pos = blockIdx.x * kBlock + threadIdx.x;
    if (pos >= n) return;
    const uint32_t seg = seg_of[pos];
    uint32_t flag = 0;
    if (seg != kNoSegment) {
        const TdNode &nd = nodes[seg];
        int axis;
        float mid, lo, hi;
        uint32_t who;
        td_axis_mid(nd, &axis, &mid);
        td_item_box(nd, idx[pos], axis, box_lo, box_hi, obj_of, obj_bounds, &lo, &hi, &who);
        flag = (hi + lo) * 0.5f > mid ? 1u : 0u;
    }
    flags[pos] = flag;
}

// One thread per segment of the level: is it one-sided (every centre on one side of the middle)?  The reference then
// splits off ONE item, the last or the first in its sorted order (true_middle stops at n - 1 or stays at 1,
// src/tracer/bvh.rs:106-117), which is how a box as large as the whole set -- a room around a scene -- leaves the
// tree at once instead of riding down with one half after the other.
__global__ void __launch_bounds__(kBlock) td_check_kernel(const uint32_t *level, uint32_t level_count, const uint32_t *flags,
                                                          const uint32_t *scan, TdNode *nodes, uint32_t *counters) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= level_count) return;
    TdNode &nd = nodes[level[t]];
    const uint32_t count = nd.end - nd.begin;
    const uint32_t n_front = scan[nd.end - 1] + flags[nd.end - 1] - scan[nd.begin];
    // (after eight such splits in a row the range is cut in the middle instead: a thousand equal boxes beside a smaller
    // one would otherwise leave one by one, a level each, as they do in the reference's recursion)
    nd.one_sided = (count > 2u && nd.peels < 8u) ? (n_front == 0u ? 1u : (n_front == count ? 2u : 0u)) : 0u;
    if (nd.one_sided) atomicAdd(&counters[2], 1u);
}
// The extreme items of the one-sided segments by the reference's sort key, "max + min / 2" on the axis (src/tracer/bvh.rs:125-126).
__global__ void __launch_bounds__(kBlock) td_extremes_kernel(uint32_t n, const uint32_t *idx, const uint32_t *seg_of, const float4 *box_lo,
                                                             const float4 *box_hi, const uint32_t *obj_of, const uint32_t *obj_bounds,
                                                             TdNode *nodes) {
    const uint32_t pos = blockIdx.x * kBlock + threadIdx.x;
    uint32_t seg = pos < n ? seg_of[pos] : kNoSegment;
    if (seg != kNoSegment && nodes[seg].one_sided == 0u) seg = kNoSegment;
    const bool valid = seg != kNoSegment;
    if (__ballot(valid) == 0ull) return;
    unsigned long long kmax = 0ull, kmin = ~0ull;
    if (valid) {
        const TdNode &nd = nodes[seg];
        int axis;
        float mid, lo, hi;
        uint32_t who;
        td_axis_mid(nd, &axis, &mid);
        td_item_box(nd, idx[pos], axis, box_lo, box_hi, obj_of, obj_bounds, &lo, &hi, &who);
        kmax = kmin = ((unsigned long long)ordered(hi + lo * 0.5f) << 32) | who;
    }
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {  // segmented over the runs of lanes of one segment, as in td_bounds_kernel
        const uint32_t other_seg = (uint32_t)__shfl_down((int)seg, off, 64);
        const unsigned long long a = __shfl_down(kmax, off, 64), b = __shfl_down(kmin, off, 64);
        if (lane + off < 64u && other_seg == seg) kmax = a > kmax ? a : kmax, kmin = b < kmin ? b : kmin;
    }
    const uint32_t before = (uint32_t)__shfl_up((int)seg, 1, 64);
    if (valid && (lane == 0u || before != seg)) atomicMax(&nodes[seg].key_max, kmax), atomicMin(&nodes[seg].key_min, kmin);
}
// New flags for the one-sided segments: the extreme item alone on its side.  Array order is the reference's visit order
// (its right child first), so the item split off at the END of the sorted order goes to the front.
__global__ void __launch_bounds__(kBlock) td_reflag_kernel(uint32_t n, const uint32_t *idx, const uint32_t *seg_of, const uint32_t *obj_of,
                                                           const TdNode *nodes, uint32_t *flags) {
    const uint32_t pos = blockIdx.x * kBlock + threadIdx.x;
    if (pos >= n) return;
    const uint32_t seg = seg_of[pos];
    if (seg == kNoSegment || nodes[seg].one_sided == 0u) return;
    const TdNode &nd = nodes[seg];
    if ((nd.key_max >> 32) == (nd.key_min >> 32)) return;  // one key for all: the split cuts the range in the middle
    const uint32_t p = idx[pos];
    const uint32_t who = nd.obj_lo != nd.obj_hi ? obj_of[p] : p;
    flags[pos] = nd.one_sided == 1u ? (who == (uint32_t)nd.key_max ? 1u : 0u) : (who == (uint32_t)nd.key_min ? 0u : 1u);
}

// One thread per segment of the level: how many of its primitives go to the front, and its two children.
__global__ void __launch_bounds__(kBlock) td_split_kernel(const uint32_t *level, uint32_t level_count, const uint32_t *flags,
                                                          const uint32_t *scan, TdNode *nodes, uint2 *children, uint32_t *parent_of_node,
                                                          uint32_t *parent_of_leaf, uint32_t *next_level, uint32_t first_new,
                                                          uint32_t *counters) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    const bool live = t < level_count;
    const uint32_t me = live ? level[t] : 0u;
    TdNode nd;
    uint32_t n_front = 0, count = 0;
    bool one_sided = false;
    if (live) {
        nd = nodes[me];
        count = nd.end - nd.begin;
        n_front = scan[nd.end - 1] + flags[nd.end - 1] - scan[nd.begin];
        one_sided = n_front == 0u || n_front == count;
        if (one_sided) n_front = count / 2u;  // (still: two primitives, or items that cannot be told apart)
        nd.n_front = n_front, nd.cut_in_middle = one_sided ? 1u : 0u;
    }
    // Binary node ids for the children that are segments of their own: one reservation per wave.  The ids of a level are
    // consecutive from first_new on, so a child's place in the next level's list is its id - first_new.
    const bool need[2] = {live && n_front > 1u, live && count - n_front > 1u};
    const unsigned long long m0 = __ballot(need[0]), m1 = __ballot(need[1]);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0u && (m0 | m1)) base = atomicAdd(&counters[0], (uint32_t)(__popcll(m0) + __popcll(m1)));
    base = (uint32_t)__shfl((int)base, 0, 64);
    const unsigned long long below = lane ? (~0ull >> (64u - lane)) : 0ull;
    const uint32_t id[2] = {base + (uint32_t)__popcll(m0 & below), base + (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1 & below)};
    if (!live) return;
    uint32_t ref[2];
    for (int side = 0; side < 2; side++) {
        const uint32_t b = side == 0 ? nd.begin : nd.begin + n_front, e = side == 0 ? nd.begin + n_front : nd.end;
        uint32_t child = kNoSegment;
        if (!need[side]) {
            ref[side] = kLeafBit | b;
            parent_of_leaf[b] = me;
        } else {
            child = id[side];
            TdNode c;
            c.begin = b, c.end = e, c.n_front = 0, c.cut_in_middle = 0, c.front = c.back = kNoSegment;
            for (int k = 0; k < 3; k++) c.lo[k] = 0xFFFFFFFFu, c.hi[k] = 0u;
            c.obj_lo = 0xFFFFFFFFu, c.obj_hi = 0u, c.one_sided = 0, c.key_max = 0ull, c.key_min = ~0ull;
            c.peels = (nd.one_sided != 0u && !one_sided) ? nd.peels + 1u : 0u;
            nodes[child] = c;
            parent_of_node[child] = me;
            next_level[child - first_new] = child;
            ref[side] = child;
        }
        if (side == 0) nd.front = child; else nd.back = child;
    }
    nodes[me] = nd;
    children[me] = make_uint2(ref[0], ref[1]);
}

__global__ void __launch_bounds__(kBlock) td_scatter_kernel(uint32_t n, const uint32_t *idx, const uint32_t *seg_of, const uint32_t *flags,
                                                            const uint32_t *scan, const TdNode *nodes, uint32_t *idx_out,
                                                            uint32_t *seg_out) {
    const uint32_t pos = blockIdx.x * kBlock + threadIdx.x;
    if (pos >= n) return;
    const uint32_t seg = seg_of[pos];
    if (seg == kNoSegment) {
        idx_out[pos] = idx[pos];
        seg_out[pos] = kNoSegment;
        return;
    }
    const TdNode &nd = nodes[seg];
    const uint32_t r = pos - nd.begin;
    uint32_t to = pos;
    bool front;
    if (nd.cut_in_middle) {
        front = r < nd.n_front;
    } else {
        const uint32_t f = scan[pos] - scan[nd.begin];  // front primitives before this one in its segment
        front = flags[pos] != 0u;
        to = front ? nd.begin + f : nd.begin + nd.n_front + (r - f);
    }
    idx_out[to] = idx[pos];
    seg_out[to] = front ? nd.front : nd.back;
}

// ---- 5. boxes bottom-up + the collapse table ----
// Per internal node b:  node_lo/node_hi = its box;  cost[b] = (root, d2, d3, d4) with
//   dj   = cheapest summed half area of wide-node root boxes below b when b's two subtrees share at most j slots
//   root = half_area(b) + d4: b is the root of a wide node of its own
// and split[b] packs, for j = 2, 3, 4, how many of the j slots go to the left child (2 bits each).
__device__ __forceinline__ bool is_terminal(uint2 ch) { return (ch.x & ch.y & kLeafBit) != 0; }

__device__ __forceinline__ float subtree_cost(uint32_t c, int k, const uint2 *children, const float4 *cost) {
    if (c & kLeafBit) return 0.0f;
    if (is_terminal(children[c])) return 0.0f;
    const float4 t = cost[c];
    float best = t.x;
    if (k >= 2) best = fminf(best, t.y);
    if (k >= 3) best = fminf(best, t.z);
    if (k >= 4) best = fminf(best, t.w);
    return best;
}

// Box of binary node `cur` from its children's and its row of the collapse table.
__device__ __forceinline__ void fit_node(uint32_t cur, const uint2 *children, const float4 *leaf_lo, const float4 *leaf_hi,
                                         float4 *node_lo, float4 *node_hi, float4 *cost, uint32_t *split) {
    const uint2 ch = children[cur];
    float4 lo[2], hi[2];
    const uint32_t c2[2] = {ch.x, ch.y};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (c2[k] & kLeafBit)
            lo[k] = leaf_lo[c2[k] & ~kLeafBit], hi[k] = leaf_hi[c2[k] & ~kLeafBit];
        else
            lo[k] = node_lo[c2[k]], hi[k] = node_hi[c2[k]];
    }
    float4 blo = make_float4(fmin_(lo[0].x, lo[1].x), fmin_(lo[0].y, lo[1].y), fmin_(lo[0].z, lo[1].z), 0.0f);
    float4 bhi = make_float4(fmax_(hi[0].x, hi[1].x), fmax_(hi[0].y, hi[1].y), fmax_(hi[0].z, hi[1].z), 0.0f);
    if (is_terminal(ch)) {
        // the box that guards a pair of primitives: Aabb::surrounding_box pads by EPSILON (src/tracer/cube.rs:179-193)
        blo.x -= kEps, blo.y -= kEps, blo.z -= kEps;
        bhi.x += kEps, bhi.y += kEps, bhi.z += kEps;
    }
    node_lo[cur] = blo;
    node_hi[cur] = bhi;
    float d[3];
    uint32_t packed = 0;
#pragma unroll
    for (int j = 2; j <= 4; j++) {
        float best = __int_as_float(0x7f800000);
        int best_k = 1;
        for (int k = 1; k < j; k++) {
            const float v = subtree_cost(ch.x, k, children, cost) + subtree_cost(ch.y, j - k, children, cost);
            if (v < best) best = v, best_k = k;
        }
        d[j - 2] = best;
        packed |= (uint32_t)best_k << (2 * (j - 2));
    }
    const float dx = bhi.x - blo.x, dy = bhi.y - blo.y, dz = bhi.z - blo.z;
    float area = dx * dy + dy * dz + dz * dx;
    if (!(area >= 0.0f)) area = 0.0f;  // a NaN box must not poison the table
    cost[cur] = make_float4(area + d[2], d[0], d[1], d[2]);
    split[cur] = packed;
}

__global__ void __launch_bounds__(kBlock) fit_kernel(int n, const uint2 *children, const uint32_t *parent_of_node,
                                                     const uint32_t *parent_of_leaf, const float4 *leaf_lo, const float4 *leaf_hi,
                                                     float4 *node_lo, float4 *node_hi, float4 *cost, uint32_t *split,
                                                     uint32_t *arrived) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t cur = parent_of_leaf[i];
    for (;;) {
        __threadfence();  // what this thread wrote for the node below is visible before it is counted
        const uint32_t before = atomicAdd(&arrived[cur], 1u);
        if (before == 0) return;  // the sibling subtree is not done yet: its thread carries on
        __threadfence();          // ... and what the sibling's thread wrote is read from memory, not from a stale line
        fit_node(cur, children, leaf_lo, leaf_hi, node_lo, node_hi, cost, split);
        if (cur == 0) return;
        cur = parent_of_node[cur];
    }
}

// The same for a tree whose nodes were made level by level (the median splits): the nodes [first, first + count) of one
// level, after the levels below it.  No counters, no fences: a launch sees what the launches before it wrote.
__global__ void __launch_bounds__(kBlock) fit_level_kernel(uint32_t first, uint32_t count, const uint2 *children, const float4 *leaf_lo,
                                                           const float4 *leaf_hi, float4 *node_lo, float4 *node_hi, float4 *cost,
                                                           uint32_t *split) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t < count) fit_node(first + t, children, leaf_lo, leaf_hi, node_lo, node_hi, cost, split);
}

// ---- 6. collapse, one level of the wide tree per launch ----
struct WideSlot {
    uint32_t word;   // child word for leaves; binary node id for internal slots
    uint32_t inner;  // 1: `word` is a binary node that roots a wide node of the next level
    float4 lo, hi;
};

// Slots of the wide node rooted at binary node b (not a terminal): b's two subtrees in at most 4 slots by the table.
__device__ __forceinline__ int wide_slots(uint32_t b, const uint2 *children, const float4 *cost, const uint32_t *split,
                                          const float4 *leaf_lo, const float4 *leaf_hi, const float4 *node_lo,
                                          const float4 *node_hi, bool greedy, WideSlot out[4]) {
    int n_out = 0;
    if (greedy) {
        // dissolve the child with the largest box first (the simpler rule; kept for comparison)
        uint32_t c[4];
        int n = 2;
        const uint2 ch = children[b];
        c[0] = ch.x, c[1] = ch.y;
        for (;;) {
            int best = -1;
            float best_area = -1.0f;
            for (int i = 0; i < n; i++) {
                if ((c[i] & kLeafBit) || is_terminal(children[c[i]])) continue;
                const float4 l = node_lo[c[i]], h = node_hi[c[i]];
                const float dx = h.x - l.x, dy = h.y - l.y, dz = h.z - l.z;
                const float a = dx * dy + dy * dz + dz * dx;
                if (a > best_area) best_area = a, best = i;
            }
            if (best < 0 || n >= 4) break;
            const uint2 k = children[c[best]];
            for (int i = n; i > best + 1; i--) c[i] = c[i - 1];  // keep left-to-right order
            c[best] = k.x, c[best + 1] = k.y;
            n++;
        }
        for (int i = 0; i < n; i++) {
            WideSlot &o = out[n_out++];
            if (c[i] & kLeafBit) {
                const uint32_t leaf = c[i] & ~kLeafBit;
                o.word = RT_CHILD_LEAF | leaf, o.inner = 0, o.lo = leaf_lo[leaf], o.hi = leaf_hi[leaf];
            } else {
                const uint2 k = children[c[i]];
                o.lo = node_lo[c[i]], o.hi = node_hi[c[i]];
                if (is_terminal(k))
                    o.word = RT_CHILD_LEAF | (1u << 28) | (k.x & ~kLeafBit), o.inner = 0;
                else
                    o.word = c[i], o.inner = 1;
            }
        }
        return n_out;
    }
    // (node, slots it may use) pairs still to be placed, left to right
    uint32_t st_node[6];
    int st_k[6];
    int sp = 0;
    {
        const uint2 ch = children[b];
        const int k = (int)((split[b] >> 4) & 3u);  // j = 4
        st_node[sp] = ch.y, st_k[sp] = 4 - k, sp++;
        st_node[sp] = ch.x, st_k[sp] = k, sp++;
    }
    while (sp > 0) {
        sp--;
        const uint32_t c = st_node[sp];
        const int k = st_k[sp];
        if (k <= 0) continue;
        WideSlot &o = out[n_out];
        if (c & kLeafBit) {
            const uint32_t leaf = c & ~kLeafBit;
            o.word = RT_CHILD_LEAF | leaf, o.inner = 0, o.lo = leaf_lo[leaf], o.hi = leaf_hi[leaf];
            n_out++;
            continue;
        }
        const uint2 ch = children[c];
        if (is_terminal(ch)) {
            o.word = RT_CHILD_LEAF | (1u << 28) | (ch.x & ~kLeafBit), o.inner = 0, o.lo = node_lo[c], o.hi = node_hi[c];
            n_out++;
            continue;
        }
        const float4 t = cost[c];
        float best = t.x;
        int best_j = 1;
        if (k >= 2 && t.y < best) best = t.y, best_j = 2;
        if (k >= 3 && t.z < best) best = t.z, best_j = 3;
        if (k >= 4 && t.w < best) best = t.w, best_j = 4;
        if (best_j == 1) {
            o.word = c, o.inner = 1, o.lo = node_lo[c], o.hi = node_hi[c];
            n_out++;
            continue;
        }
        const int kl = (int)((split[c] >> (2 * (best_j - 2))) & 3u);
        st_node[sp] = ch.y, st_k[sp] = best_j - kl, sp++;
        st_node[sp] = ch.x, st_k[sp] = kl, sp++;
    }
    return n_out;
}

// queue_in[0 .. count) = binary nodes whose wide nodes are level_begin + position; their inner children get the wide
// indices that `alloc` hands out from level_begin + count on, and are written to queue_out at (index - level_begin - count).
__global__ void __launch_bounds__(kBlock) collapse_kernel(const uint32_t *queue_in, uint32_t count, uint32_t level_begin,
                                                          uint32_t *queue_out, uint32_t *alloc, const uint2 *children,
                                                          const float4 *cost, const uint32_t *split, const float4 *leaf_lo,
                                                          const float4 *leaf_hi, const float4 *node_lo, const float4 *node_hi,
                                                          int greedy, float4 *wide) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    WideSlot slots[4];
    int n_slots = 0;
    uint32_t inner = 0;
    if (t < count) {
        n_slots = wide_slots(queue_in[t], children, cost, split, leaf_lo, leaf_hi, node_lo, node_hi, greedy != 0, slots);
        for (int i = 0; i < n_slots; i++) inner += slots[i].inner;
    }
    // exclusive prefix sum of `inner` over the block, one reservation per block
    __shared__ uint32_t wave_sum[kBlock / 64];
    __shared__ uint32_t block_base;
    uint32_t incl = inner;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += v;
    }
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < kBlock / 64; w++) {
            const uint32_t v = wave_sum[w];
            wave_sum[w] = total;
            total += v;
        }
        block_base = total ? atomicAdd(alloc, total) : 0u;
    }
    __syncthreads();
    if (t >= count) return;
    uint32_t next = block_base + wave_sum[threadIdx.x >> 6] + (incl - inner);
    float bx[6][4];
    uint32_t words[4] = {RT_CHILD_NONE, RT_CHILD_NONE, RT_CHILD_NONE, RT_CHILD_NONE};
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int i = 0; i < 4; i++) bx[a][i] = __int_as_float(0x7f800000);  // an empty slot: a box no ray hits
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i >= n_slots) continue;
        bx[0][i] = slots[i].lo.x, bx[1][i] = slots[i].lo.y, bx[2][i] = slots[i].lo.z;
        bx[3][i] = slots[i].hi.x, bx[4][i] = slots[i].hi.y, bx[5][i] = slots[i].hi.z;
        if (slots[i].inner) {
            words[i] = next;
            queue_out[next - (level_begin + count)] = slots[i].word;
            next++;
        } else {
            words[i] = slots[i].word;
        }
    }
    float4 *dst = wide + (size_t)(level_begin + t) * RT_NODE_FLOAT4S;
#pragma unroll
    for (int a = 0; a < 6; a++) dst[a] = make_float4(bx[a][0], bx[a][1], bx[a][2], bx[a][3]);
    dst[6] = make_float4(__uint_as_float(words[0]), __uint_as_float(words[1]), __uint_as_float(words[2]), __uint_as_float(words[3]));
    dst[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// the whole tree of a scene with one or two primitives: one wide node, one leaf slot
__global__ void tiny_tree_kernel(int n, const float4 *leaf_lo, const float4 *leaf_hi, float4 *node_lo, float4 *node_hi,
                                 float4 *wide) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float4 lo = leaf_lo[0], hi = leaf_hi[0];
    if (n == 2) {
        const float4 l1 = leaf_lo[1], h1 = leaf_hi[1];
        lo = make_float4(fmin_(lo.x, l1.x) - kEps, fmin_(lo.y, l1.y) - kEps, fmin_(lo.z, l1.z) - kEps, 0.0f);
        hi = make_float4(fmax_(hi.x, h1.x) + kEps, fmax_(hi.y, h1.y) + kEps, fmax_(hi.z, h1.z) + kEps, 0.0f);
    }
    node_lo[0] = lo, node_hi[0] = hi;
    const float e = __int_as_float(0x7f800000);  // the empty slots: boxes no ray hits
    wide[0] = make_float4(lo.x, e, e, e), wide[1] = make_float4(lo.y, e, e, e), wide[2] = make_float4(lo.z, e, e, e);
    wide[3] = make_float4(hi.x, e, e, e), wide[4] = make_float4(hi.y, e, e, e), wide[5] = make_float4(hi.z, e, e, e);
    wide[6] = make_float4(__uint_as_float(RT_CHILD_LEAF | ((uint32_t)(n - 1) << 28)), __uint_as_float(RT_CHILD_NONE),
                          __uint_as_float(RT_CHILD_NONE), __uint_as_float(RT_CHILD_NONE));
    wide[7] = make_float4(0, 0, 0, 0);
}

#define GB_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) {                                                   \
            err = std::string(#call) + ": " + hipGetErrorString(e_);              \
            return RT_ERR_DEVICE;                                                 \
        }                                                                         \
    } while (0)

// One device allocation cut into 256-byte aligned pieces (the build's temporaries).
struct Arena {
    char *base = nullptr;
    size_t size = 0, used = 0;
    ~Arena() {
        if (base) (void)hipFree(base);
    }
    static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
    template <class T>
    T *take(size_t count) {
        T *p = (T *)(base + used);
        used += pad(count * sizeof(T));
        return p;
    }
};

// Host arrays go to a device pool in the order they are added; small ones are gathered in a staging vector so that
// a scene of thousands of small meshes costs a few copies, large ones are copied straight from the caller's memory.
struct PoolUpload {
    struct Piece {
        const void *src;
        size_t bytes, offset;
    };
    std::vector<Piece> pieces;
    size_t total = 0;
    size_t add(const void *src, size_t bytes) {
        const size_t off = total;
        pieces.push_back(Piece{src, bytes, off});
        total += bytes;
        return off;
    }
    int run(char *dst, std::string &err) const {
        constexpr size_t kSmall = 4u << 20, kFlush = 64u << 20;
        std::vector<char> stage;
        size_t stage_off = 0;
        auto flush = [&]() -> int {
            if (stage.empty()) return RT_OK;
            GB_HIP(hipMemcpy(dst + stage_off, stage.data(), stage.size(), hipMemcpyHostToDevice));
            stage.clear();
            return RT_OK;
        };
        for (const Piece &p : pieces) {
            if (p.bytes == 0) continue;
            if (p.bytes >= kSmall) {
                int rc = flush();
                if (rc) return rc;
                GB_HIP(hipMemcpy(dst + p.offset, p.src, p.bytes, hipMemcpyHostToDevice));
                continue;
            }
            if (stage.empty()) stage_off = p.offset;
            stage.insert(stage.end(), (const char *)p.src, (const char *)p.src + p.bytes);
            if (stage.size() >= kFlush) {
                int rc = flush();
                if (rc) return rc;
            }
        }
        return flush();
    }
};

struct PtrKey {
    const void *a, *b;
    uint64_t n;
    bool operator==(const PtrKey &o) const { return a == o.a && b == o.b && n == o.n; }
};
struct PtrKeyHash {
    size_t operator()(const PtrKey &k) const {
        return std::hash<const void *>()(k.a) ^ (std::hash<const void *>()(k.b) * 1000003u) ^ (size_t)(k.n * 0x9E3779B97F4A7C15ull);
    }
};

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

void DeviceBuild::release() {
    if (nodes) (void)hipFree(nodes);
    if (prims) (void)hipFree(prims);
    if (shade_records) (void)hipFree(shade_records);
    nodes = prims = shade_records = nullptr;
}

int build_device_scene(const RtSceneDesc *d, DeviceBuild &out, std::string &err) {
    const auto t_start = std::chrono::steady_clock::now();
    const bool debug = std::getenv("RTCORE_DEBUG_TREE") != nullptr;
    uint64_t total = 0;
    {
        int rc = check_scene_desc(d, total, err, /*check_indices=*/false);  // vertex indices are checked on the device
        if (rc) return rc;
    }
    const uint32_t n = (uint32_t)total;

    // ---- sources: one segment per object; mesh arrays that several objects share are uploaded once
    std::vector<uint32_t> seg_first(d->n_objects), seg_ref(d->n_objects);
    std::vector<MeshDev> meshes;
    std::vector<SimpleDev> simples;
    PoolUpload pos_pool, nrm_pool, idx_pool;
    std::unordered_map<PtrKey, uint64_t, PtrKeyHash> vtx_seen, idx_seen;
    uint64_t id = 0, n_triangles = 0;
    for (uint64_t i = 0; i < d->n_objects; i++) {
        const RtObject &o = d->objects[i];
        seg_first[i] = (uint32_t)id;
        if (o.type == RT_OBJ_MESH) {
            const RtMesh &m = d->meshes[o.mesh];
            MeshDev md;
            std::memset(&md, 0, sizeof(md));
            mesh_rotor(m, md.rot);
            for (int k = 0; k < 3; k++) md.scale[k] = m.scale[k], md.tr[k] = m.translation[k];
            md.meta = RT_PRIM_TRIANGLE | (m.cull_backface ? 0u : RT_PRIM_FLAG) | ((uint32_t)o.material << RT_PRIM_MAT_SHIFT);
            md.n_vertices = (uint32_t)std::min<uint64_t>(m.n_vertices, 0xFFFFFFFFull);
            const PtrKey vk{m.positions, m.normals, m.n_vertices}, ik{m.indices, nullptr, m.n_faces};
            auto vit = vtx_seen.find(vk);
            if (vit == vtx_seen.end()) {
                const uint64_t off = pos_pool.add(m.positions, m.n_vertices * 12) / 12;
                nrm_pool.add(m.normals, m.n_vertices * 12);
                vit = vtx_seen.emplace(vk, off).first;
            }
            auto iit = idx_seen.find(ik);
            if (iit == idx_seen.end()) iit = idx_seen.emplace(ik, idx_pool.add(m.indices, m.n_faces * 12) / 12).first;
            md.vtx_off = vit->second;
            md.face_off = iit->second;
            seg_ref[i] = kLeafBit | (uint32_t)meshes.size();
            meshes.push_back(md);
            id += m.n_faces;
            n_triangles += m.n_faces;
        } else {
            SimpleDev sd;
            std::memset(&sd, 0, sizeof(sd));
            float box[6];
            simple_object_record(o, (uint32_t)id, sd.rec, sd.nrm, box);
            for (int k = 0; k < 3; k++) sd.box[k] = box[k], sd.box[4 + k] = box[3 + k];
            seg_ref[i] = (uint32_t)simples.size();
            simples.push_back(sd);
            if (o.type == RT_OBJ_TRIANGLE) n_triangles++;
            id += 1;
        }
    }

    // ---- device memory: outputs of their own, everything else from one arena
    DeviceBuild B;
    struct Guard {  // frees the outputs unless the build completes
        DeviceBuild *b;
        ~Guard() {
            if (b) b->release();
        }
    } guard{&B};
    B.prims_bytes = (size_t)n * 48 + 128;  // (spare bytes: see the traversal's record fetches)
    B.shade_bytes = (size_t)n * 96 + 128;
    GB_HIP(hipMalloc(&B.prims, B.prims_bytes));
    GB_HIP(hipMalloc(&B.shade_records, B.shade_bytes));

    size_t sort_tmp_bytes = 0;
    GB_HIP(rocprim::radix_sort_pairs(nullptr, sort_tmp_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                     (uint32_t *)nullptr, n, 0, 63, (hipStream_t)0));
    const char *tree_env = experiment_env("RTCORE_DEVICE_TREE");
    const bool morton_tree = tree_env && std::strcmp(tree_env, "morton") == 0;  // the Morton-order tree instead of the median splits
    {
        size_t scan_bytes = 0;
        GB_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, n, rocprim::plus<uint32_t>(),
                                       (hipStream_t)0));
        sort_tmp_bytes = std::max(sort_tmp_bytes, scan_bytes);
    }
    const size_t n_inner = n > 1 ? n - 1 : 1;
    const size_t max_wide = n_inner;
    Arena A;
    {
        auto P = Arena::pad;
        A.size = P(seg_first.size() * 4) * 2 + P(meshes.size() * sizeof(MeshDev)) + P(simples.size() * sizeof(SimpleDev)) +
                 P(pos_pool.total) + P(nrm_pool.total) + P(idx_pool.total) + 2 * P((size_t)n * 16) /*boxes*/ +
                 2 * P((size_t)n * 8) + 2 * P((size_t)n * 4) /*keys, vals*/ + P(sort_tmp_bytes) + 2 * P((size_t)n * 16) /*leaf boxes*/ +
                 P(n_inner * 8) + 2 * P((size_t)n * 4) /*children, parents*/ + 3 * P(n_inner * 16) /*node boxes, cost*/ +
                 2 * P(n_inner * 4) /*split, arrived*/ + 2 * P(n_inner * 4) /*queues*/ + P(max_wide * 128) + P(n_inner * sizeof(TdNode)) + P((size_t)n * 4) + P(seg_first.size() * 24) + P(256) + 4096;
        GB_HIP(hipMalloc((void **)&A.base, A.size));
    }
    uint32_t *d_seg_first = A.take<uint32_t>(seg_first.size()), *d_seg_ref = A.take<uint32_t>(seg_ref.size());
    MeshDev *d_meshes = A.take<MeshDev>(meshes.size());
    SimpleDev *d_simples = A.take<SimpleDev>(simples.size());
    char *d_pos = A.take<char>(pos_pool.total), *d_nrm = A.take<char>(nrm_pool.total), *d_idx = A.take<char>(idx_pool.total);
    float4 *box_lo = A.take<float4>(n), *box_hi = A.take<float4>(n);
    uint64_t *keys_a = A.take<uint64_t>(n), *keys_b = A.take<uint64_t>(n);
    uint32_t *vals_a = A.take<uint32_t>(n), *vals_b = A.take<uint32_t>(n);
    char *sort_tmp = A.take<char>(sort_tmp_bytes);
    float4 *leaf_lo = A.take<float4>(n), *leaf_hi = A.take<float4>(n);
    uint2 *children = A.take<uint2>(n_inner);
    uint32_t *parent_of_node = A.take<uint32_t>(n), *parent_of_leaf = A.take<uint32_t>(n);
    float4 *node_lo = A.take<float4>(n_inner), *node_hi = A.take<float4>(n_inner), *cost = A.take<float4>(n_inner);
    uint32_t *split = A.take<uint32_t>(n_inner), *arrived = A.take<uint32_t>(n_inner);
    uint32_t *queue_a = A.take<uint32_t>(n_inner), *queue_b = A.take<uint32_t>(n_inner);
    float4 *wide = A.take<float4>(max_wide * 8);
    TdNode *td_nodes = A.take<TdNode>(n_inner);
    uint32_t *obj_of = A.take<uint32_t>(n), *obj_bounds = A.take<uint32_t>(seg_first.size() * 6);
    uint32_t *scalars = A.take<uint32_t>(64);  // [0..5] bounds, [8] status, [9] wide-node counter, [10] binary nodes, [11] segments of the next level, [12] one-sided segments of this level
    if (A.used > A.size) {
        err = "internal: build arena too small";
        return RT_ERR_INTERNAL;
    }

    // ---- uploads
    GB_HIP(hipMemcpy(d_seg_first, seg_first.data(), seg_first.size() * 4, hipMemcpyHostToDevice));
    GB_HIP(hipMemcpy(d_seg_ref, seg_ref.data(), seg_ref.size() * 4, hipMemcpyHostToDevice));
    if (!meshes.empty()) GB_HIP(hipMemcpy(d_meshes, meshes.data(), meshes.size() * sizeof(MeshDev), hipMemcpyHostToDevice));
    if (!simples.empty()) GB_HIP(hipMemcpy(d_simples, simples.data(), simples.size() * sizeof(SimpleDev), hipMemcpyHostToDevice));
    int rc;
    if ((rc = pos_pool.run(d_pos, err)) || (rc = nrm_pool.run(d_nrm, err)) || (rc = idx_pool.run(d_idx, err))) return rc;
    {
        uint32_t init[64];
        std::memset(init, 0, sizeof(init));
        for (int k = 0; k < 3; k++) init[k] = 0xFFFFFFFFu, init[3 + k] = 0u;
        GB_HIP(hipMemcpy(scalars, init, sizeof(init), hipMemcpyHostToDevice));
    }
    {
        std::vector<uint32_t> init(seg_first.size() * 6);
        for (size_t o = 0; o < seg_first.size(); o++)
            for (int k = 0; k < 3; k++) init[6 * o + k] = 0xFFFFFFFFu, init[6 * o + 3 + k] = 0u;
        GB_HIP(hipMemcpy(obj_bounds, init.data(), init.size() * 4, hipMemcpyHostToDevice));
    }
    const double ms_upload = ms_since(t_start);

    // ---- device stages
    Sources S{d_seg_first, d_seg_ref, (uint32_t)seg_first.size(), d_meshes, d_simples, (const float *)d_pos, (const float *)d_nrm,
              (const uint32_t *)d_idx};
    const hipStream_t stream = 0;
    const uint32_t blocks_n = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(boxes_kernel, dim3(blocks_n), dim3(kBlock), 0, stream, S, n, box_lo, box_hi, scalars, scalars + 8, obj_of,
                       obj_bounds);
    GB_HIP(hipGetLastError());
    const uint32_t *perm = nullptr;  // position in the primitive array -> primitive id
    std::vector<uint32_t> level_first;  // median splits: first binary node of each level (ids are handed out level by level)
    bool hierarchy_done = false;
    if (morton_tree || n <= 2) {
        hipLaunchKernelGGL(morton_kernel, dim3(blocks_n), dim3(kBlock), 0, stream, box_lo, box_hi, n, scalars, keys_a, vals_a);
        GB_HIP(hipGetLastError());
        GB_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_tmp_bytes, keys_a, keys_b, vals_a, vals_b, n, 0, 63, stream));
        perm = vals_b;
    } else {
        // median splits, one level of the binary tree per turn
        uint32_t *idx_in = vals_a, *idx_out = vals_b;
        uint32_t *seg_in = (uint32_t *)keys_a, *seg_out = (uint32_t *)keys_a + n;
        uint32_t *flags = (uint32_t *)keys_b, *scan = (uint32_t *)keys_b + n;
        uint32_t *level_in = queue_a, *level_out = queue_b;
        hipLaunchKernelGGL(td_init_kernel, dim3(blocks_n), dim3(kBlock), 0, stream, n, idx_in, seg_in, td_nodes);
        const uint32_t start[3] = {1u, 0u, 0u};  // node 0 is the root; no segment of the next level, no one-sided segment yet
        const uint32_t zero = 0;
        GB_HIP(hipMemcpyAsync(scalars + 10, start, 12, hipMemcpyHostToDevice, stream));
        GB_HIP(hipMemcpyAsync(level_in, &zero, 4, hipMemcpyHostToDevice, stream));
        uint32_t level_count = 1, levels = 0, nodes_made = 1;
        level_first.assign(1, 0u);  // level 0 is node 0
        while (level_count) {
            if (++levels > 4096) {
                err = "internal: binary tree deeper than 4096 levels";
                return RT_ERR_INTERNAL;
            }
            if (level_count <= 4096u)
                hipLaunchKernelGGL(td_bounds_coarse_kernel, dim3((n + kBlock * kTdCoarse - 1) / (kBlock * kTdCoarse)), dim3(kBlock), 0,
                                   stream, n, idx_in, seg_in, box_lo, box_hi, obj_of, td_nodes);
            else
                hipLaunchKernelGGL(td_bounds_kernel, dim3(blocks_n), dim3(kBlock), 0, stream, n, idx_in, seg_in, box_lo, box_hi, obj_of,
                                   td_nodes);
            hipLaunchKernelGGL(td_classify_kernel, dim3(blocks_n), dim3(kBlock), 0, stream, n, idx_in, seg_in, box_lo, box_hi, obj_of,
                               obj_bounds, td_nodes, flags);
            GB_HIP(hipGetLastError());
            size_t tmp_bytes = sort_tmp_bytes;
            GB_HIP(rocprim::exclusive_scan(sort_tmp, tmp_bytes, flags, scan, 0u, n, rocprim::plus<uint32_t>(), stream));
            hipLaunchKernelGGL(td_check_kernel, dim3((level_count + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, level_in, level_count,
                               flags, scan, td_nodes, scalars + 10);
            GB_HIP(hipGetLastError());
            uint32_t n_one_sided = 0;
            GB_HIP(hipMemcpy(&n_one_sided, scalars + 12, 4, hipMemcpyDeviceToHost));  // (synchronises)
            if (n_one_sided) {
                hipLaunchKernelGGL(td_extremes_kernel, dim3(blocks_n), dim3(kBlock), 0, stream, n, idx_in, seg_in, box_lo, box_hi, obj_of,
                                   obj_bounds, td_nodes);
                hipLaunchKernelGGL(td_reflag_kernel, dim3(blocks_n), dim3(kBlock), 0, stream, n, idx_in, seg_in, obj_of, td_nodes, flags);
                GB_HIP(hipGetLastError());
                tmp_bytes = sort_tmp_bytes;
                GB_HIP(rocprim::exclusive_scan(sort_tmp, tmp_bytes, flags, scan, 0u, n, rocprim::plus<uint32_t>(), stream));
                GB_HIP(hipMemcpyAsync(scalars + 12, &zero, 4, hipMemcpyHostToDevice, stream));
            }
            hipLaunchKernelGGL(td_split_kernel, dim3((level_count + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, level_in, level_count,
                               flags, scan, td_nodes, children, parent_of_node, parent_of_leaf, level_out, nodes_made,
                               scalars + 10);
            hipLaunchKernelGGL(td_scatter_kernel, dim3(blocks_n), dim3(kBlock), 0, stream, n, idx_in, seg_in, flags, scan, td_nodes,
                               idx_out, seg_out);
            GB_HIP(hipGetLastError());
            uint32_t made_now = 0;
            GB_HIP(hipMemcpy(&made_now, scalars + 10, 4, hipMemcpyDeviceToHost));  // (synchronises)
            if (made_now < nodes_made || made_now > n_inner) {
                err = "internal: binary-node counter out of range";
                return RT_ERR_INTERNAL;
            }
            level_first.push_back(nodes_made);  // the nodes made for the next level: [nodes_made, made_now)
            level_count = made_now - nodes_made;
            nodes_made = made_now;
            std::swap(idx_in, idx_out), std::swap(seg_in, seg_out), std::swap(level_in, level_out);
        }
        perm = idx_in;
        hierarchy_done = true;
        if (debug) std::fprintf(stderr, "[rtcore] device build: %u levels of median splits\n", levels);
    }
    hipLaunchKernelGGL(emit_kernel, dim3(blocks_n), dim3(kBlock), 0, stream, S, n, perm, box_lo, box_hi, (float4 *)B.prims,
                       (float4 *)B.shade_records, leaf_lo, leaf_hi);
    GB_HIP(hipGetLastError());

    uint32_t n_wide = 0, depth = 0;
    const char *collapse = experiment_env("RTCORE_COLLAPSE");
    const int greedy = (collapse && std::strcmp(collapse, "greedy") == 0) ? 1 : 0;
    if (n <= 2) {
        hipLaunchKernelGGL(tiny_tree_kernel, dim3(1), dim3(64), 0, stream, (int)n, leaf_lo, leaf_hi, node_lo, node_hi, wide);
        GB_HIP(hipGetLastError());
        n_wide = 1, depth = 1;
    } else {
        GB_HIP(hipMemsetAsync(arrived, 0, n_inner * 4, stream));
        if (!hierarchy_done)
            hipLaunchKernelGGL(hierarchy_kernel, dim3((n - 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, keys_b, (int)n, children,
                               parent_of_node, parent_of_leaf);
        if (hierarchy_done) {
            level_first.push_back((uint32_t)n_inner);  // (end of the last level)
            for (size_t lv = level_first.size() - 1; lv-- > 0;) {
                const uint32_t first = level_first[lv], cnt = level_first[lv + 1] - first;
                if (cnt)
                    hipLaunchKernelGGL(fit_level_kernel, dim3((cnt + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, first, cnt, children,
                                       leaf_lo, leaf_hi, node_lo, node_hi, cost, split);
            }
        } else {
            hipLaunchKernelGGL(fit_kernel, dim3(blocks_n), dim3(kBlock), 0, stream, (int)n, children, parent_of_node, parent_of_leaf,
                               leaf_lo, leaf_hi, node_lo, node_hi, cost, split, arrived);
        }
        GB_HIP(hipGetLastError());
        // level 0 is the root (binary node 0; not a terminal because n > 2)
        const uint32_t zero = 0, one = 1;
        GB_HIP(hipMemcpyAsync(queue_a, &zero, 4, hipMemcpyHostToDevice, stream));
        GB_HIP(hipMemcpyAsync(scalars + 9, &one, 4, hipMemcpyHostToDevice, stream));
        uint32_t level_begin = 0, count = 1;
        uint32_t *qin = queue_a, *qout = queue_b;
        while (count) {
            if (depth > 4096) {
                err = "internal: wide tree deeper than 4096 levels";
                return RT_ERR_INTERNAL;
            }
            hipLaunchKernelGGL(collapse_kernel, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, qin, count, level_begin,
                               qout, scalars + 9, children, cost, split, leaf_lo, leaf_hi, node_lo, node_hi, greedy, wide);
            GB_HIP(hipGetLastError());
            uint32_t allocated = 0;
            GB_HIP(hipMemcpy(&allocated, scalars + 9, 4, hipMemcpyDeviceToHost));  // (synchronises)
            depth++;
            level_begin += count;
            if (allocated < level_begin || allocated > max_wide) {
                err = "internal: wide-node counter out of range";
                return RT_ERR_INTERNAL;
            }
            count = allocated - level_begin;
            std::swap(qin, qout);
        }
        n_wide = level_begin;
    }
    uint32_t host_scalars[16];
    float4 root_box[2];
    GB_HIP(hipMemcpy(host_scalars, scalars, sizeof(host_scalars), hipMemcpyDeviceToHost));
    GB_HIP(hipMemcpy(&root_box[0], node_lo, 16, hipMemcpyDeviceToHost));
    GB_HIP(hipMemcpy(&root_box[1], node_hi, 16, hipMemcpyDeviceToHost));
    if (host_scalars[8]) {
        err = "a mesh has a vertex index out of range";
        return RT_ERR_INVALID_ARGUMENT;
    }
    if ((uint64_t)n_wide >= (1ull << 25)) {  // kernels address nodes with 32-bit byte offsets
        err = "too many tree nodes";
        return RT_ERR_INVALID_ARGUMENT;
    }
    B.nodes_bytes = (size_t)n_wide * 128 + 128;
    GB_HIP(hipMalloc(&B.nodes, B.nodes_bytes));
    GB_HIP(hipMemcpy(B.nodes, wide, (size_t)n_wide * 128, hipMemcpyDeviceToDevice));
    GB_HIP(hipDeviceSynchronize());

    if (debug && n > 2) {  // summed half area of the binary tree's nodes
        std::vector<float4> hl(n_inner), hh(n_inner);
        GB_HIP(hipMemcpy(hl.data(), node_lo, n_inner * 16, hipMemcpyDeviceToHost));
        GB_HIP(hipMemcpy(hh.data(), node_hi, n_inner * 16, hipMemcpyDeviceToHost));
        double sum = 0.0;
        for (size_t i = 0; i < n_inner; i++) {
            double dx = hh[i].x - hl[i].x, dy = hh[i].y - hl[i].y, dz = hh[i].z - hl[i].z;
            sum += dx * dy + dy * dz + dz * dx;
        }
        std::fprintf(stderr, "[rtcore] binary tree: %zu nodes, summed half area %.6g\n", n_inner, sum);
    }
    if (debug) {  // surface-area cost of the wide tree, for comparing builders (as the host build prints it)
        std::vector<float> hn((size_t)n_wide * 32);
        GB_HIP(hipMemcpy(hn.data(), B.nodes, hn.size() * 4, hipMemcpyDeviceToHost));
        double inner = 0.0, leaf = 0.0;
        size_t n_leaf1 = 0, n_leaf2 = 0, n_inner_slots = 0;
        for (size_t nd = 0; nd < n_wide; nd++) {
            const float *f = &hn[nd * 32];
            uint32_t w[4];
            std::memcpy(w, f + 24, 16);
            for (int c = 0; c < 4; c++) {
                if (w[c] == RT_CHILD_NONE) continue;
                double dx = f[12 + c] - f[c], dy = f[16 + c] - f[4 + c], dz = f[20 + c] - f[8 + c];
                double a = dx * dy + dy * dz + dz * dx;
                if (w[c] & RT_CHILD_LEAF) { leaf += a; if ((w[c] >> 28) & 7u) n_leaf2++; else n_leaf1++; } else { inner += a; n_inner_slots++; }
            }
        }
        std::fprintf(stderr, "[rtcore] wide tree: %u nodes, summed half area of node children %.6g, of leaf children %.6g; %zu node slots, %zu single and %zu pair leaf slots\n",
                     n_wide, inner, leaf, n_inner_slots, n_leaf1, n_leaf2);
    }
    B.n_prims = n, B.n_triangles = n_triangles, B.n_nodes = n_wide, B.n_binary_nodes = n > 1 ? n - 1 : 1;
    B.wide_depth = depth;
    B.max_stack = 3 * depth + 4;
    B.root_lo[0] = root_box[0].x, B.root_lo[1] = root_box[0].y, B.root_lo[2] = root_box[0].z;
    B.root_hi[0] = root_box[1].x, B.root_hi[1] = root_box[1].y, B.root_hi[2] = root_box[1].z;
    B.ms_upload = ms_upload;
    B.ms_total = ms_since(t_start);
    if (debug)
        std::fprintf(stderr, "[rtcore] device build: %u primitives, %u wide nodes, %u levels, upload %.1f ms, total %.1f ms\n", n, n_wide,
                     depth, B.ms_upload, B.ms_total);
    out = B;
    guard.b = nullptr;
    return RT_OK;
}

}  // namespace rtcore
