// rtcore.hip -- kernels of the wavefront path tracer and the C ABI (include/rtcore.h).
//
// Stages of one batch of paths (all state is structure-of-arrays float4 in HBM):
//
//   generate    one lane per path: camera ray (Camera::gen_ray), throughput 1, RNG key
//   extend      one lane per live path: closest hit through the 4-wide tree
//   shade       one lane per live path: scatter / Russian roulette / emitter / sky;
//               finished paths store their radiance sample, survivors are compacted
//               into the other state buffer (wave ballot + prefix popcount, one
//               atomic per wave)
//   accumulate  one lane per pixel: sums that pixel's samples in sample order with
//               the reference's non-finite scrub and adds the mean to the frame
//
// A batch holds all `samples` paths of a run of consecutive pixels; a path's radiance slot is
// pixel * samples + sample, and every radiance sample is written exactly once -- no float
// atomics anywhere.
// Live-path counts stay on the device; kernels read them and idle lanes exit, so
// a pass needs no host synchronisation.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rtcore.h"
#include "rt_device.hpp"
#include "rt_traverse.hpp"
#include "scene_build.hpp"
#include "compact.hpp"
#include "gpu_build.hpp"

using namespace rtd;

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kCountSlots = 1024;  // live-path counts [0, kCountSlots), then the work counters of the extend launches
// A launch's rays are dealt out from kWorkShares counters, each in a 128-byte line of its own (see extend_kernel)
#ifndef RT_WORK_SHARES
#define RT_WORK_SHARES 16
#endif
constexpr uint32_t kWorkShares = RT_WORK_SHARES, kWorkStride = 32, kWorkWords = kWorkShares * kWorkStride;
#ifndef RT_TILE_ROWS
#define RT_TILE_ROWS 8
#endif
constexpr uint32_t kTileRows = RT_TILE_ROWS;
constexpr float kTMin = 0.00015f;   // src/render.rs:15
constexpr float kTMax = 100000.0f;  // src/render.rs:16

struct PathState {
    float4 *o;    // origin, .w = radiance slot (bits)
    float4 *d;    // direction (normalised), .w unused
    float4 *thr;  // throughput, .w = RNG key (bits)
};

// ---------------------------------------------------------------------------
// generate: Renderer::render's per-sample setup (src/render.rs:176-201)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) generate_kernel(RtDevCamera cam, uint32_t width, uint32_t height,
                                                           uint32_t spp, uint32_t pass, uint32_t seed, uint32_t p0,
                                                           uint32_t np, uint32_t preview, PathState out,
                                                           uint32_t *counts, uint32_t n_counts, uint32_t wave_samples) {
    uint32_t n = np * spp;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        counts[0] = n;
        for (uint32_t k = 1; k < n_counts; k++) counts[k] = 0;  // (the extend launches' work counters are cleared by the host's memset)
    }
    if (i >= n) return;
    // Which (sample, pixel offset) path i is.  64 consecutive paths -- a wave of the closest-hit stage, which takes
    // camera rays 64 at a time -- are `wave_samples` samples (a power of two that divides spp) of 64 / wave_samples
    // neighbouring pixels: rays through one pixel meet the same nodes and leaves all the way down, so the wave's lanes
    // stay in step.  (wave_samples = 1: one sample of 64 pixels.)
    uint32_t s, j;
    if (wave_samples > 1) {
        const uint32_t per_group = np * wave_samples;  // paths of one group of wave_samples samples
        const uint32_t g = i / per_group, r = i - g * per_group;
        const uint32_t lane = r & 63u;
        s = g * wave_samples + (lane & (wave_samples - 1u));
        j = (r >> 6) * (64u / wave_samples) + lane / wave_samples;
    } else {
        s = i / np, j = i - s * np;
    }
    // Consecutive pixels (j) are taken from 8-row strips column by column (a permutation of the batch's pixel
    // offsets), so the pixels that share a wave are neighbours down a column -- with one sample per wave an 8 x 8
    // tile of the frame rather than a 64 x 1 line: their rays meet the same nodes.  The radiance slot is
    // `offset * spp + sample` whatever the order of the paths.
    uint32_t off = j;
    const uint32_t strip = kTileRows * width, whole = np / strip * strip;
    if (j < whole) {
        const uint32_t b = j / strip, r = j - b * strip;
        off = (b * kTileRows + r % kTileRows) * width + r / kTileRows;
    }
    uint32_t pixel = p0 + off;
    float x = (float)(pixel % width);
    float y = (float)(height - 1 - pixel / width);
    float jx, jy;
    pixel_jitter(seed, pixel, pass, s, &jx, &jy);
    uint32_t key = path_key(seed, pixel, preview ? pass : pass * spp + s);
    F3 org, dir;
    gen_ray(cam, width, height, x, y, jx, jy, &org, &dir);
    dir = normalized(dir);  // Ray::new (src/ray.rs:10-15)
    out.o[i] = make_float4(org.x, org.y, org.z, __uint_as_float(off * spp + s));  // radiance slot: [pixel][sample]
    out.d[i] = make_float4(dir.x, dir.y, dir.z, 0.0f);
    out.thr[i] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(key));
}

// ---------------------------------------------------------------------------
// extend: Bvh::hit for every live path (src/tracer/bvh.rs:143)
// hit = (t, visit rank)   rank -1 = miss; the consumer completes the HitRecord (barycentrics,
// normal, material, primitive id) with finish_hit, where all lanes of a wave are busy
// ---------------------------------------------------------------------------
// Persistent waves: a wave reserves runs of kRayChunk consecutive ray indices from the counter of
// its share of the launch (see the kernel) and hands them to its lanes; a lane that has finished its ray parks its result, and
// once kServiceLanes lanes are parked or empty the wave writes the parked results and refills
// those lanes, so the lanes of a wave work on different rays at different depths instead of
// idling until the slowest ray of a fixed group of 64 is done.
#ifndef RT_RAY_CHUNK
#define RT_RAY_CHUNK 128
#endif
#ifndef RT_SERVICE_LANES
#define RT_SERVICE_LANES 24
#endif
#ifndef RT_SERVICE_LANES_WIDE
#define RT_SERVICE_LANES_WIDE 40
#endif
#ifndef RT_EXTEND_WAVES
#define RT_EXTEND_WAVES 7
#endif
#ifndef RT_VOTE_LEAF_WEIGHT
#define RT_VOTE_LEAF_WEIGHT 3
#endif
#ifndef RT_VOTE_NODE_WEIGHT
#define RT_VOTE_NODE_WEIGHT 2
#endif
constexpr uint32_t kVoteLeafWeight = RT_VOTE_LEAF_WEIGHT;
constexpr uint32_t kVoteNodeWeight = RT_VOTE_NODE_WEIGHT;
#ifndef RT_POP_TURNS
#define RT_POP_TURNS 1
#endif
constexpr int kPopTurns = RT_POP_TURNS;
constexpr uint32_t kRayChunk = RT_RAY_CHUNK;
// refill threshold: over the compact nodes / over the 128-byte nodes (whose fetches cost more, so that rays refilled
// together -- they walk the upper tree together and share its lines -- are worth more idle lanes)
constexpr uint32_t kServiceLanesCompact = RT_SERVICE_LANES, kServiceLanesWide = RT_SERVICE_LANES_WIDE;
// Camera rays are another matter: every 64 consecutive paths are a group of neighbouring rays (generate_kernel: the
// samples of a few neighbouring pixels, or one sample of an 8 x 8 tile), a run of kRayChunk rays is whole
// groups, and a wave that refills only when all its lanes are done takes exactly one group at a time -- 64 rays
// that walk the tree in step.  Measured on the camera rays alone (one bounce, 8 x 8 tiles): +7 % on the 10 M-triangle
// scene, +11 % on 1 M triangles, +3 % on the tea set, +1 % on the spheres scene, +0.3 % on Cornell; thresholds
// between 24 and 64 that split groups over waves (32, 56, 60) are slower than either, and 64 stays the best with the
// samples of a pixel in one wave.  The rays of the first bounce already want the ordinary threshold (16 and 24
// equal, 32 / 48 / 64: -2 / -4 / -21 %).
#ifndef RT_SERVICE_LANES_PRIMARY
#define RT_SERVICE_LANES_PRIMARY 64
#endif
constexpr uint32_t kServiceLanesPrimary = RT_SERVICE_LANES_PRIMARY;
static_assert(RT_RAY_CHUNK % 64 == 0, "a run of rays is whole groups of 64");

// Number of different values of `v` among the lanes with `on` (instrumented builds only: one turn per value).
__device__ __forceinline__ uint32_t wave_distinct(bool on, uint32_t v) {
    unsigned long long rest = __ballot(on);
    uint32_t n = 0;
    while (rest) {
        const uint32_t leader = (uint32_t)__shfl((int)v, __ffsll((long long)rest) - 1, 64);
        rest &= ~__ballot(on && v == leader);
        n++;
    }
    return n;
}

template <bool COUNT, bool COMPACT>
__global__ void __launch_bounds__(kBlock, RT_EXTEND_WAVES)
    extend_kernel(RtSceneView sc, const float4 *__restrict__ ro, const float4 *__restrict__ rd,
                  const uint32_t *__restrict__ count_ptr, uint32_t n_fixed, float t_min, float t_max,
                  float2 *__restrict__ hit, uint2 *ovf, uint32_t *head, unsigned long long *counters,
                  uint32_t service_lanes) {
    __shared__ uint2 s_stack[RT_LDS_STACK * kBlock];
    const uint32_t kServiceLanes = service_lanes;  // (wave-uniform: idle lanes at which the wave refills)
    const uint32_t n = count_ptr ? *count_ptr : n_fixed;
    const uint32_t lane = threadIdx.x & 63u;
    Stack st;
    st.lds = s_stack + threadIdx.x;
    st.lds_stride = kBlock;
    st.ovf = ovf;
    st.ovf_stride = gridDim.x * blockDim.x;
    st.sp = 0;
    uint32_t visits = 0, tests = 0;
    unsigned long long lookups = 0;  // (wave-uniform) line look-ups of the node and primitive fetches
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // leaf steps, lanes in them, node steps, lanes in them, lanes without a ray over all steps, refills (lane 0 of each wave)
    Traversal tr;
    // What the lane is doing -- at a node, at a leaf, taking the next stack entry, finished with its result not yet
    // written (parked), without a ray (empty) -- is ONE register, so every question the wave asks of its lanes is one
    // compare whose result is the lane mask.
    uint32_t state = kEmpty;
    int32_t ray_i = 0;                       // ray this lane works on (state < kEmpty)
    uint32_t chunk_next = 0, chunk_end = 0;  // wave-uniform: the reserved run of ray indices
    bool no_more = false;                    // wave-uniform: every share of the launch has run out
    auto lanes = [](bool p) { return (unsigned long long)__builtin_amdgcn_ballot_w64(p); };

    // Work distribution.  One global counter takes about 84 reservations per microsecond from the whole chip, which
    // runs of 256 rays turn into a ceiling of ~21 Grays/s (scenes with cheap rays sat at it: Cornell 19.4 Grays/s in
    // this stage with one counter, 30 with these) and which forbids the shorter runs that expensive rays do better
    // with.  So the launch's rays are cut into kWorkShares contiguous shares, each with a counter in a line of its
    // own; a wave starts on the share of its block's number modulo kWorkShares -- blocks are dealt to the eight XCDs
    // in turn, so a share's rays, neighbours in the frame, mostly meet in one XCD's L2 -- and moves on to the next
    // share when its own has run out (every share is drained by whoever gets there; a wave never goes back).
    // Measured: 8 / 16 / 32 shares and runs of 64 / 128 / 192 / 256 rays; 16 x 128 is the best or within 2 % of it on
    // every configuration.
    const uint32_t per_share = ((n + kWorkShares * 64u - 1u) / (kWorkShares * 64u)) * 64u;  // whole groups of 64 rays
    uint32_t share = blockIdx.x % kWorkShares, shares_left = kWorkShares;                  // wave-uniform
    const long long loop_t0 = COUNT ? clock64() : 0;
    for (;;) {
        const uint32_t n_idle = (uint32_t)__popcll(lanes(state >= kParked));
        if (n_idle >= kServiceLanes && (!no_more || lanes(state == kParked) != 0ull)) {
            if (COUNT && lane == 0) dbg[5]++;
            const long long svc_t0 = COUNT ? clock64() : 0;
            if (state == kParked) {
                hit[ray_i] = make_float2(tr.best.t, __int_as_float(tr.best.index()));
                state = kEmpty;
            }
            // refill; rays that miss the root box (or are NaN) are answered on the spot, so repeat
            while (!no_more) {
                const unsigned long long empty_mask = lanes(state == kEmpty);
                const uint32_t want = (uint32_t)__popcll(empty_mask);
                if (want < kServiceLanes) break;
                if (chunk_next == chunk_end) {
                    // a run from the share this wave is on; a share that has run out sends the wave on to the next one
                    bool got = false;
                    while (shares_left != 0u) {
                        const uint32_t lo = share * per_share < n ? share * per_share : n;
                        const uint32_t size = n - lo < per_share ? n - lo : per_share;
                        uint32_t b = 0;
                        if (lane == 0) b = atomicAdd(head + share * kWorkStride, kRayChunk);
                        b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                        if (b < size) {
                            chunk_next = lo + b;
                            chunk_end = lo + ((b + kRayChunk) < size ? (b + kRayChunk) : size);
                            got = true;
                            break;
                        }
                        share = (share + 1u) % kWorkShares;
                        shares_left--;
                    }
                    if (!got) {
                        no_more = true;
                        break;
                    }
                }
                const uint32_t avail = chunk_end - chunk_next;
                const uint32_t give = want < avail ? want : avail;
                if (state == kEmpty) {
                    // empty lanes below this one
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(empty_mask >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)empty_mask, 0u));
                    if (rank < give) {
                        const uint32_t i = chunk_next + rank;
                        float4 o4 = ro[i], d4 = rd[i];
                        tr.ray.o = f3(o4.x, o4.y, o4.z);
                        tr.ray.d = f3(d4.x, d4.y, d4.z);
                        if (traversal_begin<COUNT>(sc, tr, st, t_min, t_max, &visits)) {
                            ray_i = (int32_t)i;
                            state = kAtNode;  // node 0
                        } else {
                            hit[i] = make_float2(0.0f, __int_as_float(-1));
                        }
                    }
                }
                chunk_next += give;
            }
            if (COUNT && lane == 0) dbg[6] += (unsigned long long)(clock64() - svc_t0);
        }
        // One step for part of the wave: lanes at a node and lanes at a leaf need different code,
        // so each iteration runs only the kind that more lanes are waiting for; the others wait.
        const bool at_leaf = state == kAtLeaf, at_node = state == kAtNode;
        const uint32_t n_leaf = (uint32_t)__popcll(lanes(at_leaf));
        const uint32_t n_node = (uint32_t)__popcll(lanes(at_node));
        if (n_leaf + n_node == 0u && lanes(state == kPopping) == 0ull) {
            if (no_more && lanes(state == kParked) == 0ull) break;
            continue;
        }
        // The kind with the (weighted) majority runs.
        const bool run_leaf = n_leaf * kVoteLeafWeight >= n_node * kVoteNodeWeight;
        const bool run_node = !run_leaf;
        if (COUNT && lane == 0) {  // occupancy of the two step kinds, for tuning
            if (run_leaf && n_leaf) dbg[0]++, dbg[1] += n_leaf;
            if (run_node && n_node) dbg[2]++, dbg[3] += n_node;
            dbg[4] += 64u - n_leaf - n_node;
        }
        if (COUNT) {
            // What the step asks of the vector L1: every load instruction looks up each DIFFERENT 128-byte line its
            // lanes address.  A compact node is four loads from half a line, a leaf block five or eight loads from one
            // line; a 128-byte node is seven loads from one line and a primitive record three loads, 48 bytes that may
            // straddle two lines.  Lanes at the same node or in the same line share a look-up.
            uint32_t lk = 0;
            if (run_node && n_node) lk += COMPACT ? 4u * wave_distinct(at_node, tr.entry >> 1) : 7u * wave_distinct(at_node, tr.entry);
            if (COMPACT && run_leaf && n_leaf) {  // one block = one line: two loads for the guard, three per record
                const bool two = ((tr.entry >> 28) & 7u) != 0u;
                lk += 5u * wave_distinct(at_leaf, tr.entry & RT_LEAF_FIRST_MASK) + 3u * wave_distinct(at_leaf && two, tr.entry & RT_LEAF_FIRST_MASK);
            }
            if (!COMPACT && run_leaf && n_leaf) {
                // 16-byte word index of the leaf's first record (48 bytes a record): the line of word w is w >> 3
                const uint32_t word0 = (tr.entry & RT_LEAF_FIRST_MASK) * RT_PRIM_FLOAT4S;
                const bool two = ((tr.entry >> 28) & 7u) != 0u;
                for (uint32_t k = 0; k < 6u; k++) lk += wave_distinct(at_leaf && (k < 3u || two), (word0 + k) >> 3);
            }
            lookups += lk;
        }
        if (run_leaf) {
            if (at_leaf) {
                if (COMPACT)
                    traversal_guarded_leaf_step<COUNT>(sc, tr, t_min, t_max, &tests);
                else
                    traversal_leaf_step<COUNT>(sc, tr, t_min, t_max, &tests);
                state = kPopping;
            }
        } else if (at_node) {
            state = COMPACT ? traversal_qnode_step<COUNT>(sc, tr, st, t_min, &visits) : traversal_node_step<COUNT>(sc, tr, st, t_min, &visits);
        }
        // Next entry for the lanes that want one.  After a hit most of a ray's stack is beyond the
        // best distance and is dropped entry by entry; a wave waits for its slowest lane (an unbounded
        // loop here took 5.1 turns per leaf step and 1.5 per node step), so each step is followed by
        // kPopTurns turns of two entries each, and a lane with a longer run of dead entries goes on
        // with it on the wave's next turns (it takes no part in the steps meanwhile).
#pragma unroll
        for (int turn = 0; turn < kPopTurns; turn++)
            if (state == kPopping) state = traversal_pop_turn(tr, st);
    }
    if (COUNT) {
        atomicAdd(&counters[0], (unsigned long long)visits);
        atomicAdd(&counters[1], (unsigned long long)tests);
        if (lane == 0)
        {
            dbg[7] = (unsigned long long)(clock64() - loop_t0);
            for (int k = 0; k < 8; k++) atomicAdd(&counters[2 + k], dbg[k]);
            atomicAdd(&counters[14], lookups);
        }
    }
}

// Appends the lanes with `alive` to a queue.  One atomic per block: each wave counts its live
// lanes with a ballot, wave leaders leave the counts in LDS, one thread reserves the block's run
// and every lane takes base + (live lanes of earlier waves) + (live lanes below it in its wave).
// A single global counter sustains only about 90 atomics per microsecond, so per-wave atomics
// would cap a 4K pass at that rate.  Must be called by every thread of the block.
#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 1024
#endif
constexpr int kShadeBlock = RT_SHADE_BLOCK;
__device__ __forceinline__ uint32_t block_compact_slot(bool alive, uint32_t *counter, uint32_t *s_counts) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const unsigned long long mask = __ballot(alive);
    if (lane == 0) s_counts[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < n_waves; w++) {
            uint32_t c = s_counts[w];
            s_counts[w] = total;
            total += c;
        }
        s_counts[n_waves] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    const uint32_t dst = s_counts[n_waves] + s_counts[wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();  // s_counts is reused by the next call
    return dst;
}

// ---------------------------------------------------------------------------
// shade: the body of ray_color's loop (src/render.rs:62-81) for bounce `bounce`
// ---------------------------------------------------------------------------
// Two flags per thread, one reservation per block.
__device__ __forceinline__ void block_compact_slots2(bool alive0, bool alive1, uint32_t *counter, uint32_t *s_counts,
                                                     uint32_t *dst0, uint32_t *dst1) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const unsigned long long m0 = __ballot(alive0), m1 = __ballot(alive1);
    if (lane == 0) s_counts[wave] = (uint32_t)(__popcll(m0) + __popcll(m1));
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < n_waves; w++) {
            uint32_t c = s_counts[w];
            s_counts[w] = total;
            total += c;
        }
        s_counts[n_waves] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t base = s_counts[n_waves] + s_counts[wave];
    *dst0 = base + (uint32_t)__popcll(m0 & below);
    *dst1 = base + (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1 & below);
    __syncthreads();  // s_counts is reused by the next call
}

// One path of shade_kernel: returns whether the path goes on; if so (new_o, new_d, thr) is its next state,
// otherwise its radiance sample has been stored.
__device__ __forceinline__ bool shade_path(const RtSceneView &sc, const RtDevMaterial *mats, const PathState &in,
                                           const float2 *__restrict__ hit_a, float4 *__restrict__ rad, uint32_t i,
                                           uint32_t bounce, uint32_t depth, float light_clamp, F3 *new_o, F3 *new_d,
                                           F3 *thr_out, float *slot_bits, float *key_bits) {
    const float4 o4 = in.o[i], d4 = in.d[i], th4 = in.thr[i];
    const float2 ha = hit_a[i];
    const uint32_t slot = __float_as_uint(o4.w);
    const uint32_t key = __float_as_uint(th4.w);
    *slot_bits = o4.w;
    *key_bits = th4.w;
    Ray ray;
    ray.o = f3(o4.x, o4.y, o4.z);
    ray.d = f3(d4.x, d4.y, d4.z);
    F3 thr = f3(th4.x, th4.y, th4.z);
    F3 result;
    bool alive = false;
    if (__float_as_int(ha.y) >= 0) {
        F3 nrm;
        uint32_t mat;
        int32_t pid;
        finish_hit(sc, ray, ha.x, __float_as_int(ha.y), &nrm, &mat, &pid);
        RtDevMaterial m = mats[mat];
        F3 sdir;
        F3 att = scatter(m, ray, ha.x, nrm, key, bounce, &sdir);
        if (max3(att) <= 1.0f) {
            thr = thr * att;
            if (max3(thr) < draw(key, bounce, DIM_RR)) {
                result = thr * 0.01f;  // Russian-roulette break, src/render.rs:68-70,83
            } else {
                thr = thr * (1.0f / max3(thr));
                if (bounce + 1 == depth) {
                    result = thr * 0.01f;  // depth exhausted, src/render.rs:83
                } else {
                    alive = true;
                    *new_o = ray_at(ray, ha.x);
                    *new_d = normalized(sdir);
                }
            }
        } else {
            result = thr * clamped(att, 0.0f, 1.0f * light_clamp);  // emitter, src/render.rs:73-77
        }
    } else {
        result = thr * get_sky(sc, ray.d, light_clamp);  // src/render.rs:80
    }
    if (!alive) rad[slot] = make_float4(result.x, result.y, result.z, 0.0f);
    *thr_out = thr;
    return alive;
}

__global__ void __launch_bounds__(kShadeBlock)
    shade_kernel(RtSceneView sc, PathState in, const float2 *__restrict__ hit_a,
                 const uint32_t *__restrict__ count_in, PathState out, uint32_t *count_out, float4 *__restrict__ rad,
                 uint32_t bounce, uint32_t depth, float light_clamp) {
    __shared__ uint32_t s_counts[kShadeBlock / 64 + 1];
    const uint32_t n = *count_in;
    const RtDevMaterial *mats = (const RtDevMaterial *)sc.materials;
    // A block takes runs of 2 * kShadeBlock paths, two per thread, and reserves their places in the output
    // queue with ONE atomic (the queue counter sustains only about 90 atomics per microsecond).  Every thread of
    // a block runs the same number of turns: the compaction has barriers.
    const uint32_t run = 2u * (uint32_t)kShadeBlock;
    const uint32_t n_runs = (n + run - 1u) / run;
    for (uint32_t r = blockIdx.x; r < n_runs; r += gridDim.x) {
        const uint32_t i0 = r * run + threadIdx.x, i1 = i0 + (uint32_t)kShadeBlock;
        bool alive0 = false, alive1 = false;
        F3 o0, d0, t0, o1, d1, t1;
        float slot0 = 0.0f, key0 = 0.0f, slot1 = 0.0f, key1 = 0.0f;
        if (i0 < n) alive0 = shade_path(sc, mats, in, hit_a, rad, i0, bounce, depth, light_clamp, &o0, &d0, &t0, &slot0, &key0);
        if (i1 < n) alive1 = shade_path(sc, mats, in, hit_a, rad, i1, bounce, depth, light_clamp, &o1, &d1, &t1, &slot1, &key1);
        uint32_t dst0, dst1;
        block_compact_slots2(alive0, alive1, count_out, s_counts, &dst0, &dst1);
        if (alive0) {
            out.o[dst0] = make_float4(o0.x, o0.y, o0.z, slot0);
            out.d[dst0] = make_float4(d0.x, d0.y, d0.z, 0.0f);
            out.thr[dst0] = make_float4(t0.x, t0.y, t0.z, key0);
        }
        if (alive1) {
            out.o[dst1] = make_float4(o1.x, o1.y, o1.z, slot1);
            out.d[dst1] = make_float4(d1.x, d1.y, d1.z, 0.0f);
            out.thr[dst1] = make_float4(t1.x, t1.y, t1.z, key1);
        }
    }
}

// normals_only / colors_only / no_bounce (src/render.rs:107-138): every path ends after its first hit
__global__ void __launch_bounds__(kBlock)
    shade_first_hit_kernel(RtSceneView sc, PathState in, const float2 *__restrict__ hit_a,
                           const uint32_t *__restrict__ count_in,
                           float4 *__restrict__ rad, int mode) {
    const uint32_t n = *count_in;
    const uint32_t stride = gridDim.x * blockDim.x;
    const RtDevMaterial *mats = (const RtDevMaterial *)sc.materials;
    const float kInf = __int_as_float(0x7f800000);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float4 o4 = in.o[i], d4 = in.d[i], th4 = in.thr[i];
        float2 ha = hit_a[i];
        uint32_t slot = __float_as_uint(o4.w);
        Ray ray;
        ray.o = f3(o4.x, o4.y, o4.z);
        ray.d = f3(d4.x, d4.y, d4.z);
        F3 result;
        if (__float_as_int(ha.y) >= 0) {
            F3 nrm;
            uint32_t mat;
            int32_t pid;
            finish_hit(sc, ray, ha.x, __float_as_int(ha.y), &nrm, &mat, &pid);
            RtDevMaterial m = mats[mat];
            if (mode == RT_MODE_NORMALS) {
                result = (nrm + f3(1.0f, 1.0f, 1.0f)) * 0.5f;
            } else if (mode == RT_MODE_COLORS) {
                result = f3(m.albedo[0], m.albedo[1], m.albedo[2]);
            } else {
                F3 sdir;
                F3 att = scatter(m, ray, ha.x, nrm, __float_as_uint(th4.w), 0, &sdir);
                float k = rclamp(dot(normalized(f3(1.0f, 1.0f, -0.5f)), nrm), 0.1f, 1.0f);
                result = k * att;
            }
        } else {
            result = get_sky(sc, ray.d, kInf);
        }
        rad[slot] = make_float4(result.x, result.y, result.z, 0.0f);
    }
}

__global__ void __launch_bounds__(kBlock) fill_radiance_kernel(float4 *rad, uint32_t n, float v) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rad[i] = make_float4(v, v, v, 0.0f);
}

// ---------------------------------------------------------------------------
// accumulate: the tail of Renderer::render's per-pixel closure (src/render.rs:203-214)
// ---------------------------------------------------------------------------
// The radiance buffer is [pixel][sample] (a wave of camera rays writes runs of one pixel's samples, see
// generate_kernel), and a pixel's sum is sequential (the scrub looks at every partial sum).  One wave per block takes 64
// pixels: it loads their samples kAccumChunk per pixel at a time as contiguous runs -- every line is read once, whole --
// into LDS, and each lane then adds up its own pixel's run in sample order.  (Rounds of 4 / 8 / 16 samples measured:
// 8 is the best on the large frames, where 16 leaves too few waves on a CU; 4 wins only on the 640 x 480 one.)
#ifndef RT_ACCUM_CHUNK
#define RT_ACCUM_CHUNK 8
#endif
constexpr uint32_t kAccumChunk = RT_ACCUM_CHUNK, kAccumPitch = kAccumChunk + 1;  // (the odd pitch spreads the lanes' rows over the banks)
__global__ void __launch_bounds__(64)
    accumulate_kernel(const float4 *__restrict__ rad, uint32_t np, uint32_t spp, uint32_t p0, float *accum,
                      int overwrite) {
    __shared__ float4 tile[64 * kAccumPitch];
    const uint32_t lane = threadIdx.x, base = blockIdx.x * 64u;
    if (base >= np) return;
    const uint32_t j = base + lane;
    const uint32_t live = np - base < 64u ? np - base : 64u;  // pixels of this wave
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    const float kInf = __int_as_float(0x7f800000);
    if (!overwrite) {
        for (uint32_t c0 = 0; c0 < spp; c0 += kAccumChunk) {
            const uint32_t t = spp - c0 < kAccumChunk ? spp - c0 : kAccumChunk;
            for (uint32_t m = lane; m < live * t; m += 64u) {
                const uint32_t pp = m / t, ss = m - pp * t;
                tile[pp * kAccumPitch + ss] = rad[(size_t)(base + pp) * spp + c0 + ss];
            }
            __syncthreads();
            if (lane < live) {
                for (uint32_t ss = 0; ss < t; ss++) {
                    const float4 c = tile[lane * kAccumPitch + ss];
                    sx += c.x;
                    sy += c.y;
                    sz += c.z;
                    if (!(fabsf(sx) < kInf)) sx = 0.0f;
                    if (!(fabsf(sy) < kInf)) sy = 0.0f;
                    if (!(fabsf(sz) < kInf)) sz = 0.0f;
                }
            }
            __syncthreads();
        }
    }
    if (j >= np) return;
    size_t p = (size_t)(p0 + j) * 3;
    if (overwrite) {
        // Renderer::preview returns the sample itself (src/render.rs:243-259)
        float4 c = rad[j];
        accum[p] = c.x, accum[p + 1] = c.y, accum[p + 2] = c.z;
    } else {
        float f = (float)spp;
        accum[p] = accum[p] + sx / f;
        accum[p + 1] = accum[p + 1] + sy / f;
        accum[p + 2] = accum[p + 2] + sz / f;
    }
}

__global__ void tally_kernel(const uint32_t *counts, uint32_t n_bounces, unsigned long long *totals) {
    unsigned long long rays = 0;
    for (uint32_t b = 0; b < n_bounces; b++) rays += counts[b];
    totals[0] += rays;
    totals[1] += counts[0];
}

// ---------------------------------------------------------------------------
// closest-hit batch I/O
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    pack_rays_kernel(const float *__restrict__ org_dir, uint32_t n, float4 *ro, float4 *rd) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *r = org_dir + (size_t)i * 6;
    F3 d = normalized(f3(r[3], r[4], r[5]));  // Ray::new
    ro[i] = make_float4(r[0], r[1], r[2], 0.0f);
    rd[i] = make_float4(d.x, d.y, d.z, 0.0f);
}
__global__ void __launch_bounds__(kBlock)
    unpack_hits_kernel(RtSceneView sc, const float2 *__restrict__ hit_a, const float4 *__restrict__ ro,
                       const float4 *__restrict__ rd, uint32_t n, int32_t *prim_id, float *t, float *normal,
                       float *point) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 ha = hit_a[i];
    float4 o4 = ro[i], d4 = rd[i];
    bool ok = __float_as_int(ha.y) >= 0;
    Ray ray;
    ray.o = f3(o4.x, o4.y, o4.z);
    ray.d = f3(d4.x, d4.y, d4.z);
    F3 nrm = f3(0.0f, 0.0f, 0.0f);
    uint32_t mat = 0;
    int32_t pid = -1;
    if (ok) finish_hit(sc, ray, ha.x, __float_as_int(ha.y), &nrm, &mat, &pid);
    if (prim_id) prim_id[i] = pid;
    if (t) t[i] = ok ? ha.x : 0.0f;
    if (normal) {
        normal[(size_t)i * 3] = nrm.x, normal[(size_t)i * 3 + 1] = nrm.y, normal[(size_t)i * 3 + 2] = nrm.z;
    }
    if (point) {
        F3 p = ok ? ray_at(ray, ha.x) : f3(0.0f, 0.0f, 0.0f);
        point[(size_t)i * 3] = p.x, point[(size_t)i * 3 + 1] = p.y, point[(size_t)i * 3 + 2] = p.z;
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
thread_local std::string g_error;

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
#define RT_HIP(call)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(RT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));               \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }  // a scene that fails half-way through its uploads frees what it already holds
    int ensure(size_t want) {
        if (want <= bytes) return RT_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        if (hipMalloc(&p, want) != hipSuccess) {
            p = nullptr;
            return fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc of " + std::to_string(want) + " bytes failed");
        }
        bytes = want;
        return RT_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct TimedSpan {
    hipEvent_t a, b;
    int stage;  // 0 extend, 1 shade, 2 other
};

}  // namespace

struct RtScene {
    int device = 0;
    int n_cus = 256;
    rtcore::HostScene host;  // metadata only after upload (big arrays are released)
    DevBuf nodes, qnodes, leaves, prims, shade_records, materials, env;
    bool compact = false;  // the closest-hit kernel walks the compact nodes (RTCORE_NODES=compact)
    RtSceneView view{};
    // work buffers
    DevBuf st_o[2], st_d[2], st_thr[2], hit_a, rad, counts, totals, ovf, counters, accum_tmp, io_a, io_b, io_c,
        io_d, io_e;
    uint64_t work_paths = 0;
    uint64_t paths_in_flight = 0;  // default batch size, fixed at the first call that needs it
    uint32_t extend_blocks = 0;
    bool timing = false;
    bool counting = false;
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    RtStats stats{};
    uint64_t device_bytes = 0;
    int builder = RT_BUILD_EXACT;  // which build made the tree
    double build_ms = 0.0;         // host clock of that build (replicas: of the one build they share)
};

namespace {

// persistent waves: a launch is as many blocks as are resident at once
uint32_t resident_extend_blocks(int n_cus, bool compact) {
    int per_cu = 0;
    const hipError_t e = compact ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, extend_kernel<false, true>, kBlock, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, extend_kernel<false, false>, kBlock, 0);
    if (e != hipSuccess || per_cu < 1)
        per_cu = 4;
    return (uint32_t)n_cus * (uint32_t)per_cu;
}

int ensure_work(RtScene *s, uint64_t paths) {
    if (paths <= s->work_paths && s->extend_blocks) return RT_OK;
    int rc;
    for (int k = 0; k < 2; k++) {
        if ((rc = s->st_o[k].ensure(paths * 16))) return rc;
        if ((rc = s->st_d[k].ensure(paths * 16))) return rc;
        if ((rc = s->st_thr[k].ensure(paths * 16))) return rc;
    }
    if ((rc = s->hit_a.ensure(paths * 8))) return rc;
    if ((rc = s->rad.ensure(paths * 16))) return rc;
    if ((rc = s->counts.ensure(((size_t)kCountSlots + (size_t)kCountSlots * kWorkWords) * sizeof(uint32_t)))) return rc;
    if (!s->totals.p) {
        if ((rc = s->totals.ensure(64))) return rc;
        RT_HIP(hipMemset(s->totals.p, 0, 64));
    }
    if (!s->counters.p) {
        if ((rc = s->counters.ensure(128))) return rc;
        RT_HIP(hipMemset(s->counters.p, 0, 128));
    }
    if (!s->extend_blocks) s->extend_blocks = resident_extend_blocks(s->n_cus, s->compact);
    uint32_t deep = s->view.max_stack > RT_LDS_STACK ? s->view.max_stack - RT_LDS_STACK : 0;
    size_t ovf_bytes = (size_t)deep * s->extend_blocks * kBlock * sizeof(uint2);
    if (ovf_bytes) {
        if ((rc = s->ovf.ensure(ovf_bytes))) return rc;
    }
    s->work_paths = paths;
    return RT_OK;
}

hipEvent_t take_event(RtScene *s) {
    if (s->events_used == s->event_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        s->event_pool.push_back(e);
    }
    return s->event_pool[s->events_used++];
}
struct StageTimer {
    RtScene *s;
    hipStream_t stream;
    hipEvent_t a = nullptr;
    int stage;
    StageTimer(RtScene *sc, hipStream_t st, int stg) : s(sc), stream(st), stage(stg) {
        if (s->timing) {
            a = take_event(s);
            if (a) (void)hipEventRecord(a, stream);
        }
    }
    ~StageTimer() {
        if (s->timing && a) {
            hipEvent_t b = take_event(s);
            if (b) {
                (void)hipEventRecord(b, stream);
                s->spans.push_back(TimedSpan{a, b, stage});
            }
        }
    }
};

uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }

// `head` points at the kWorkWords words of this launch's work counters; they must be zero when the kernel starts.
// `coherent`: the rays are camera rays in the order generate_kernel makes them (see kServiceLanesPrimary).
int launch_extend(RtScene *s, hipStream_t stream, const float4 *ro, const float4 *rd, const uint32_t *count_ptr,
                  uint32_t n_upper, float t_min, float t_max, bool counted, uint32_t *head, bool coherent = false) {
    if (n_upper == 0) return RT_OK;
    const uint32_t service_lanes = coherent ? kServiceLanesPrimary : (s->compact ? kServiceLanesCompact : kServiceLanesWide);
    uint32_t blocks = std::min(blocks_for(n_upper), s->extend_blocks);
    StageTimer tm(s, stream, 0);
    auto kernel = counted ? (s->compact ? extend_kernel<true, true> : extend_kernel<true, false>)
                          : (s->compact ? extend_kernel<false, true> : extend_kernel<false, false>);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBlock), 0, stream, s->view, ro, rd, count_ptr, n_upper, t_min, t_max,
                       (float2 *)s->hit_a.p, (uint2 *)s->ovf.p, head, (unsigned long long *)s->counters.p, service_lanes);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

// ---- work memory of the entry point that allocates nothing (rt_closest_hit_batch_noalloc_device) ----
// One array per GPU, held from the moment the library's code is loaded on that GPU until the process ends.
// Layout: [0, kScratchRo) the launch's work counters; packed ray origins, packed directions, hits for kNoAllocMaxRays
// rays; the rest is the overflow area of the traversal stacks.  In host code the array's NAME is not a device
// address: the address on each device comes from hipGetSymbolAddress (scratch_base) and only that goes to kernels.
constexpr size_t kNoAllocMaxRays = 1u << 18;
constexpr size_t kScratchBytes = 64ull << 20;
constexpr size_t kScratchRo = kWorkWords * sizeof(uint32_t), kScratchRd = kScratchRo + kNoAllocMaxRays * 16, kScratchHit = kScratchRd + kNoAllocMaxRays * 16,
                 kScratchOvf = kScratchHit + kNoAllocMaxRays * 8;
static_assert(kScratchOvf % 256 == 0 && kScratchOvf < kScratchBytes, "scratch layout");
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
void *g_scratch_base[kMaxDevices];  // per device: each GPU holds its own copy of the array at its own address
std::mutex g_scratch_mutex;

// The array's address on `device`, which must be the current device.  Resolved once per device.
int scratch_base(int device, unsigned char **out) {
    if (device < 0 || device >= kMaxDevices) return fail(RT_ERR_DEVICE, "device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_base[device]) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            return fail(RT_ERR_DEVICE, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_base[device] = ptr;
    }
    *out = (unsigned char *)g_scratch_base[device];
    return RT_OK;
}

int check_params(const RtScene *s, const RtRenderParams *p, const void *buf) {
    if (!s || !p || !buf) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (p->width < 2 || p->height < 2) return fail(RT_ERR_INVALID_ARGUMENT, "width and height must be at least 2");
    if ((uint64_t)p->width * p->height > 0x7fffffffull) return fail(RT_ERR_INVALID_ARGUMENT, "frame too large");
    if (p->samples == 0) return fail(RT_ERR_INVALID_ARGUMENT, "samples must be positive");
    if (p->max_bounce > 1000) return fail(RT_ERR_INVALID_ARGUMENT, "max_bounce too large");
    if (p->mode < RT_MODE_NORMALS || p->mode > RT_MODE_IMAGE) return fail(RT_ERR_INVALID_ARGUMENT, "unknown mode");
    if (p->pixel_count && (p->first_pixel >= (uint64_t)p->width * p->height ||
                           p->pixel_count > (uint64_t)p->width * p->height - p->first_pixel))
        return fail(RT_ERR_INVALID_ARGUMENT, "pixel range outside the frame");
    return RT_OK;
}

// Paths (or rays) in flight per launch when the caller leaves the choice to the core: up to 128 M
// (128 B of work state each), at most a quarter of the free HBM, at least 1 M.  Asked of the device
// once per scene (hipMemGetInfo synchronises with the host), then kept.
uint64_t default_paths_in_flight(RtScene *s) {
    if (s->paths_in_flight) return s->paths_in_flight;
    uint64_t budget = 1ull << 27;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        uint64_t fit = (uint64_t)(free_b + s->work_paths * 128ull) / 4ull / 128ull;
        if (fit < (1ull << 20)) fit = 1ull << 20;
        if (budget > fit) budget = fit;
    }
    s->paths_in_flight = budget;
    return budget;
}

// Samples of one pixel that share a wave of camera rays (see generate_kernel): the largest power of two up to
// RT_WAVE_SAMPLES that divides spp, provided the batch's pixels fill whole waves that way.
#ifndef RT_WAVE_SAMPLES
#define RT_WAVE_SAMPLES 64
#endif
uint32_t wave_samples_for(uint32_t spp, uint32_t np) {
    uint32_t ws = 1;
    while (ws * 2 <= (uint32_t)RT_WAVE_SAMPLES && spp % (ws * 2) == 0) ws *= 2;
    while (ws > 1 && np % (64u / ws) != 0) ws /= 2;
    return ws;
}

// One pass (or one preview frame) on device memory.
int render_device(RtScene *s, const RtRenderParams *p, float *d_accum, hipStream_t stream, bool preview) {
    int rc = check_params(s, p, d_accum);
    if (rc) return rc;
    RT_HIP(hipSetDevice(s->device));
    const uint32_t spp = preview ? 1u : p->samples;
    const uint64_t first = p->pixel_count ? p->first_pixel : 0;
    const uint64_t pixels = p->pixel_count ? p->pixel_count : (uint64_t)p->width * p->height;  // pixels to render
    // Paths in flight per batch.  Every bounce is one launch over the batch's survivors, so small
    // batches leave the late bounces with too few rays to fill the chip and pay a tail per launch
    // (10 M-triangle scene at 4K: 1.35 / 1.70 / 1.96 / 2.12 Grays/s at 4 / 8 / 16 / 32 M paths).
    // Default: the whole pass, up to 128 M paths (128 B of state each) or a quarter of the free HBM.
    uint64_t budget = p->batch_paths;
    if (budget == 0) budget = default_paths_in_flight(s);
    if (budget > (1ull << 30)) budget = 1ull << 30;
    uint64_t np_max = budget / spp;
    if (np_max == 0) np_max = 1;
    if (np_max > pixels) np_max = pixels;
    if (np_max * spp > 0x7fffffffull) return fail(RT_ERR_INVALID_ARGUMENT, "samples too large for one batch");
    if ((rc = ensure_work(s, np_max * spp))) return rc;

    RtDevCamera cam;
    rtcore::make_camera(s->host.camera, (float)p->width / (float)p->height, cam);
    const bool first_hit_only = preview || p->mode != RT_MODE_IMAGE;
    const uint32_t depth = first_hit_only ? 1u : p->max_bounce;
    const uint32_t n_counts = depth + 2;
    if (n_counts > kCountSlots) return fail(RT_ERR_INVALID_ARGUMENT, "max_bounce too large");
    uint32_t *counts = (uint32_t *)s->counts.p;
    uint32_t *work = counts + kCountSlots;  // kWorkWords per extend launch
    PathState st[2];
    for (int k = 0; k < 2; k++) st[k] = PathState{(float4 *)s->st_o[k].p, (float4 *)s->st_d[k].p, (float4 *)s->st_thr[k].p};
    float4 *rad = (float4 *)s->rad.p;

    for (uint64_t p0 = first; p0 < first + pixels; p0 += np_max) {
        const uint32_t np = (uint32_t)std::min<uint64_t>(np_max, first + pixels - p0);
        const uint32_t n = np * spp;
        {
            StageTimer tm(s, stream, 2);
            if (depth) RT_HIP(hipMemsetAsync(work, 0, (size_t)depth * kWorkWords * sizeof(uint32_t), stream));
            hipLaunchKernelGGL(generate_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, cam, p->width,
                               p->height, spp, p->pass_index, p->seed, (uint32_t)p0, np, preview ? 1u : 0u, st[0],
                               counts, n_counts, wave_samples_for(spp, np));
        }
        if (depth == 0) {
            // ray_color's loop body never runs: every sample is 1 * 0.01 (src/render.rs:61,83)
            hipLaunchKernelGGL(fill_radiance_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, rad, n, 0.01f);
        }
        int cur = 0;
        for (uint32_t b = 0; b < depth; b++) {
            if ((rc = launch_extend(s, stream, st[cur].o, st[cur].d, counts + b, n, kTMin, kTMax, s->counting,
                                    work + (size_t)b * kWorkWords, /*coherent=*/b == 0)))
                return rc;
            StageTimer tm(s, stream, 1);
            uint32_t blocks = std::min(blocks_for(n), (uint32_t)s->n_cus * 16u);
            if (first_hit_only) {
                hipLaunchKernelGGL(shade_first_hit_kernel, dim3(blocks), dim3(kBlock), 0, stream, s->view, st[cur],
                                   (const float2 *)s->hit_a.p, counts + b, rad,
                                   preview && p->mode == RT_MODE_IMAGE ? (int)RT_MODE_IMAGE : (int)p->mode);
            } else {
                uint32_t sblocks = std::min((n + 2 * kShadeBlock - 1) / (2 * kShadeBlock), (uint32_t)s->n_cus * 4u);
                hipLaunchKernelGGL(shade_kernel, dim3(sblocks), dim3(kShadeBlock), 0, stream, s->view, st[cur],
                                   (const float2 *)s->hit_a.p, counts + b, st[cur ^ 1],
                                   counts + b + 1, rad, b, depth, p->light_clamp);
            }
            cur ^= 1;
        }
        {
            StageTimer tm(s, stream, 2);
            hipLaunchKernelGGL(accumulate_kernel, dim3((np + 63u) / 64u), dim3(64), 0, stream, rad, np, spp,
                               (uint32_t)p0, d_accum, preview ? 1 : 0);
            hipLaunchKernelGGL(tally_kernel, dim3(1), dim3(1), 0, stream, counts, depth, (unsigned long long *)s->totals.p);
        }
        RT_HIP(hipGetLastError());
    }
    return RT_OK;
}

void collect_timing(RtScene *s) {
    for (auto &sp : s->spans) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            if (sp.stage == 0) {
                s->stats.extend_ms += ms;
                s->stats.extend_launches += 1;
            }
            else if (sp.stage == 1)
                s->stats.shade_ms += ms;
            else
                s->stats.other_ms += ms;
        }
    }
    s->spans.clear();
    s->events_used = 0;
}

}  // namespace

// lets the other translation units of the library report through rt_last_error()
extern "C" int rt_set_error_message(int code, const char *msg) { return fail(code, msg ? msg : ""); }

extern "C" {

const char *rt_last_error(void) { return g_error.c_str(); }
const char *rt_version(void) {
#ifdef RTCORE_EXPERIMENTS
    return "rtcore-mi355x 0.3 (gfx950) +experiments";
#else
    return "rtcore-mi355x 0.3 (gfx950)";
#endif
}

int rt_init(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(RT_ERR_DEVICE, "no HIP device is visible");
    if (device >= 0) {
        if (device >= n) return fail(RT_ERR_INVALID_ARGUMENT, "device index out of range");
        RT_HIP(hipSetDevice(device));
    }
    int cur = 0;
    RT_HIP(hipGetDevice(&cur));
    unsigned char *base = nullptr;
    return scratch_base(cur, &base);  // the no-allocation entry point's work memory on this device
}

// Spreads the builders' back-to-back shade records out to RT_SHADE_RECORD_FLOAT4S words each.
__global__ void __launch_bounds__(kBlock) spread_shade_records_kernel(const float4 *__restrict__ in, size_t n_words, float4 *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    const size_t rec = i / (2 * RT_PRIM_FLOAT4S), w = i % (2 * RT_PRIM_FLOAT4S);
    out[rec * RT_SHADE_RECORD_FLOAT4S + w] = in[i];
}

// Puts a built scene on the CURRENT device.  `h` keeps its arrays (several replicas may be made from one build).
// The tree and the primitive records come from `h` (host build), from `dev` (device build on this device; the
// scene takes the allocations over) or from `peer` (a scene on another or the same device: device-to-device copies).
static int upload_scene(const rtcore::HostScene &h, rtcore::DeviceBuild *dev, const RtScene *peer, RtScene **out) {
    std::unique_ptr<RtScene> s(new RtScene());
    // metadata only: the device copies are what the kernels read
    s->host.env_w = h.env_w, s->host.env_h = h.env_h;
    for (int k = 0; k < 3; k++) s->host.root_lo[k] = h.root_lo[k], s->host.root_hi[k] = h.root_hi[k];
    s->host.n_prims = h.n_prims, s->host.n_triangles = h.n_triangles, s->host.n_nodes = h.n_nodes;
    s->host.n_binary_nodes = h.n_binary_nodes, s->host.wide_depth = h.wide_depth, s->host.max_stack = h.max_stack;
    s->host.camera = h.camera;
    RT_HIP(hipGetDevice(&s->device));
    hipDeviceProp_t prop;
    RT_HIP(hipGetDeviceProperties(&prop, s->device));
    s->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // RTCORE_NODES=wide keeps the 128-byte nodes (exact child boxes) for the closest-hit kernel
    s->compact = true;
    if (const char *e = std::getenv("RTCORE_NODES")) s->compact = std::string(e) != "wide";
    if (peer) s->compact = peer->compact;
    s->extend_blocks = resident_extend_blocks(s->n_cus, s->compact);
    int rc;
    auto upload = [&](DevBuf &b, const void *src, size_t bytes) -> int {
        if (bytes == 0) return RT_OK;
        int r = b.ensure(bytes + 128);  // 128 spare bytes, see ensure_work
        if (r) return r;
        RT_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        return RT_OK;
    };
    if (dev) {
        s->nodes.p = dev->nodes, s->nodes.bytes = dev->nodes_bytes;
        s->prims.p = dev->prims, s->prims.bytes = dev->prims_bytes;
        s->shade_records.p = dev->shade_records, s->shade_records.bytes = dev->shade_bytes;
        dev->nodes = dev->prims = dev->shade_records = nullptr;
    } else if (peer) {
        auto copy = [&](DevBuf &b, const DevBuf &from) -> int {
            if (!from.p) return RT_OK;
            int r = b.ensure(from.bytes);
            if (r) return r;
            RT_HIP(hipMemcpyPeer(b.p, s->device, from.p, peer->device, from.bytes));
            return RT_OK;
        };
        if ((rc = copy(s->nodes, peer->nodes)) || (rc = copy(s->qnodes, peer->qnodes)) || (rc = copy(s->leaves, peer->leaves)) ||
            (rc = copy(s->prims, peer->prims)) || (rc = copy(s->shade_records, peer->shade_records)))
            return rc;
    } else {
        if ((rc = upload(s->nodes, h.nodes.data(), h.nodes.size() * 4))) return rc;
        if ((rc = upload(s->prims, h.prims.data(), h.prims.size() * 4))) return rc;
        if ((rc = upload(s->shade_records, h.shade_records.data(), h.shade_records.size() * 4))) return rc;
    }
    if ((rc = upload(s->materials, h.materials.data(), h.materials.size() * sizeof(RtDevMaterial)))) return rc;
    if ((rc = upload(s->env, h.env.data(), h.env.size() * 4))) return rc;
    if (s->compact && !peer) {
        // the closest-hit kernel reads compact nodes and leaf blocks; the 128-byte nodes and the primitive records they
        // are made from go (a tree whose leaf blocks a child word cannot address keeps the 128-byte nodes)
        rtcore::CompactTree ct;
        std::string err;
        rc = rtcore::compact_tree(s->nodes.p, h.n_nodes, s->prims.p, h.n_prims, ct, err);
        if (rc == RT_ERR_INVALID_ARGUMENT) {
            s->compact = false;
            s->extend_blocks = resident_extend_blocks(s->n_cus, false);
        } else if (rc != RT_OK) {
            return fail(rc, err);
        } else {
            s->qnodes.p = ct.qnodes, s->qnodes.bytes = ct.qnodes_bytes;
            s->leaves.p = ct.leaves, s->leaves.bytes = ct.leaves_bytes;
            s->nodes.release();
            s->prims.release();
        }
    }
    if (RT_SHADE_RECORD_FLOAT4S != 2 * RT_PRIM_FLOAT4S && !peer && h.n_prims) {
        DevBuf wide;
        if ((rc = wide.ensure((size_t)h.n_prims * RT_SHADE_RECORD_FLOAT4S * 16 + 128))) return rc;
        RT_HIP(hipMemset(wide.p, 0, wide.bytes));
        const size_t n_words = (size_t)h.n_prims * 2 * RT_PRIM_FLOAT4S;
        hipLaunchKernelGGL(spread_shade_records_kernel, dim3((unsigned)((n_words + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0,
                           (const float4 *)s->shade_records.p, n_words, (float4 *)wide.p);
        RT_HIP(hipGetLastError());
        RT_HIP(hipDeviceSynchronize());
        s->shade_records.release();
        s->shade_records.p = wide.p, s->shade_records.bytes = wide.bytes;
        wide.p = nullptr, wide.bytes = 0;
    }
    for (const DevBuf *b : {&s->nodes, &s->qnodes, &s->leaves, &s->prims, &s->shade_records, &s->materials, &s->env})
        s->device_bytes += b->bytes;
    RtSceneView &v = s->view;
    v.nodes = s->nodes.p;
    v.qnodes = s->qnodes.p;
    v.leaves = s->leaves.p;
    v.prims = s->prims.p;
    v.shade_records = s->shade_records.p;
    v.materials = s->materials.p;
    v.env = (const float *)s->env.p;
    v.env_w = h.env_w;
    v.env_h = h.env_h;
    v.n_nodes = (uint32_t)h.n_nodes;
    v.n_prims = (uint32_t)h.n_prims;
    for (int k = 0; k < 3; k++) v.root_lo[k] = h.root_lo[k], v.root_hi[k] = h.root_hi[k];
    v.max_stack = h.max_stack;
    *out = s.release();
    return RT_OK;
}

// RT_BUILD_DEFAULT: the environment decides (RTCORE_BUILD=gpu selects the device build), else the exact host build.
static int resolve_builder(int builder) {
    if (builder == RT_BUILD_DEFAULT) {
        const char *e = std::getenv("RTCORE_BUILD");
        return (e && (std::strcmp(e, "gpu") == 0 || std::strcmp(e, "device") == 0)) ? RT_BUILD_DEVICE : RT_BUILD_EXACT;
    }
    return builder;
}

int rt_scene_create(const RtSceneDesc *desc, RtScene **out) { return rt_scene_create_with(desc, RT_BUILD_DEFAULT, out); }

int rt_scene_create_with(const RtSceneDesc *desc, int builder, RtScene **out) {
    int dev = 0;
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return fail(RT_ERR_DEVICE, "no HIP device is visible; the core has no CPU path");
    RT_HIP(hipGetDevice(&dev));
    return rt_scene_create_replicas_with(desc, builder, &dev, 1, out);
}

int rt_scene_create_replicas(const RtSceneDesc *desc, const int *devices, int n, RtScene **out) {
    return rt_scene_create_replicas_with(desc, RT_BUILD_DEFAULT, devices, n, out);
}

// One build, one replica per entry of `devices` (a device may be named more than once).  The host build is uploaded
// to every device; the device build runs on devices[0] and the other replicas are device-to-device copies of it.
int rt_scene_create_replicas_with(const RtSceneDesc *desc, int builder, const int *devices, int n, RtScene **out) {
    if (!desc || !out || !devices || n < 1) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    for (int i = 0; i < n; i++) out[i] = nullptr;
    if (builder != RT_BUILD_DEFAULT && builder != RT_BUILD_EXACT && builder != RT_BUILD_DEVICE)
        return fail(RT_ERR_INVALID_ARGUMENT, "unknown builder");
    builder = resolve_builder(builder);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return fail(RT_ERR_DEVICE, "no HIP device is visible; the core has no CPU path");
    for (int i = 0; i < n; i++)
        if (devices[i] < 0 || devices[i] >= n_dev) return fail(RT_ERR_INVALID_ARGUMENT, "device index out of range");
    int before = 0;
    RT_HIP(hipGetDevice(&before));
    const auto t0 = std::chrono::steady_clock::now();
    rtcore::HostScene h;
    rtcore::DeviceBuild db;
    std::string err;
    int rc;
    if (builder == RT_BUILD_DEVICE) {
        if (hipSetDevice(devices[0]) != hipSuccess) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
        rc = rtcore::fill_materials_camera_env(desc, h, err);
        if (rc == RT_OK) rc = rtcore::build_device_scene(desc, db, err);
        if (rc) {
            (void)hipSetDevice(before);
            return fail(rc, err);
        }
        h.n_prims = db.n_prims, h.n_triangles = db.n_triangles, h.n_nodes = db.n_nodes, h.n_binary_nodes = db.n_binary_nodes;
        h.wide_depth = db.wide_depth, h.max_stack = db.max_stack;
        for (int k = 0; k < 3; k++) h.root_lo[k] = db.root_lo[k], h.root_hi[k] = db.root_hi[k];
    } else {
        rc = rtcore::build_host_scene(desc, h, err);
        if (rc) return fail(rc, err);
    }
    const double build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < n && rc == RT_OK; i++) {
        if (hipSetDevice(devices[i]) != hipSuccess) rc = fail(RT_ERR_DEVICE, "hipSetDevice failed");
        if (rc == RT_OK)
            rc = upload_scene(h, (builder == RT_BUILD_DEVICE && i == 0) ? &db : nullptr,
                              (builder == RT_BUILD_DEVICE && i > 0) ? out[0] : nullptr, &out[i]);
        if (rc == RT_OK) out[i]->builder = builder, out[i]->build_ms = build_ms;
    }
    db.release();  // (only holds something when the first upload failed)
    (void)hipSetDevice(before);
    if (rc != RT_OK) {
        const std::string msg = g_error;
        for (int i = 0; i < n; i++) {
            if (out[i]) rt_scene_destroy(out[i]);
            out[i] = nullptr;
        }
        return fail(rc, msg);
    }
    return RT_OK;
}

int rt_scene_validate(const RtSceneDesc *desc, RtSceneInfo *info) {
    if (!desc || !info) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    rtcore::HostScene h;
    std::string err;
    int rc = rtcore::build_host_scene(desc, h, err);
    if (rc) return fail(rc, err);
    std::memset(info, 0, sizeof(*info));
    info->n_primitives = h.n_prims;
    info->n_triangles = h.n_triangles;
    info->n_wide_nodes = h.n_nodes;
    info->n_binary_nodes = h.n_binary_nodes;
    info->wide_depth = h.wide_depth;
    // what rt_scene_create would hold on the device for this build under the current environment: compact nodes and
    // leaf blocks (the default), or the 128-byte nodes and the primitive records (RTCORE_NODES=wide, or a tree whose
    // leaf blocks a child word cannot address), plus the records the shading stage reads, materials
    // and environment; every array carries 128 spare bytes
    uint64_t n_blocks = 0;
    for (size_t nd = 0; nd < h.n_nodes; nd++) {
        uint32_t w[4];
        std::memcpy(w, &h.nodes[nd * 32 + 24], 16);
        for (int c = 0; c < 4; c++)
            if (w[c] != RT_CHILD_NONE && (w[c] & RT_CHILD_LEAF)) n_blocks += (((w[c] >> 28) & 7u) + 2u) >> 1;
    }
    bool compact = true;
    if (const char *e = std::getenv("RTCORE_NODES")) compact = std::string(e) != "wide";
    if (n_blocks > (uint64_t)RT_LEAF_FIRST_MASK) compact = false;
    auto padded = [](uint64_t bytes) { return bytes ? bytes + 128 : 0; };
    info->node_bytes = (compact ? RT_QNODE_FLOAT4S : RT_NODE_FLOAT4S) * 16;
    info->primitive_bytes = RT_PRIM_FLOAT4S * 16;
    info->device_bytes = (compact ? padded(h.n_nodes * (uint64_t)(RT_QNODE_FLOAT4S * 16)) + padded(n_blocks * (uint64_t)(RT_LEAF_BLOCK_FLOAT4S * 16))
                                  : padded(h.n_nodes * (uint64_t)(RT_NODE_FLOAT4S * 16)) + padded(h.n_prims * (uint64_t)(RT_PRIM_FLOAT4S * 16))) +
                         padded(h.n_prims * (uint64_t)(RT_SHADE_RECORD_FLOAT4S * 16)) + padded(h.env.size() * 4) + padded(h.materials.size() * sizeof(RtDevMaterial));
    info->builder = RT_BUILD_EXACT;
    for (int k = 0; k < 3; k++) info->bounds_min[k] = h.root_lo[k], info->bounds_max[k] = h.root_hi[k];
    return RT_OK;
}

int rt_scene_destroy(RtScene *s) {
    if (!s) return RT_OK;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    for (hipEvent_t e : s->event_pool) (void)hipEventDestroy(e);
    delete s;  // the buffers free themselves
    return RT_OK;
}

int rt_scene_info(const RtScene *s, RtSceneInfo *info) {
    if (!s || !info) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    std::memset(info, 0, sizeof(*info));
    info->n_primitives = s->host.n_prims;
    info->n_triangles = s->host.n_triangles;
    info->n_wide_nodes = s->host.n_nodes;
    info->n_binary_nodes = s->host.n_binary_nodes;
    info->wide_depth = s->host.wide_depth;
    info->node_bytes = (s->compact ? RT_QNODE_FLOAT4S : RT_NODE_FLOAT4S) * 16;
    info->primitive_bytes = RT_PRIM_FLOAT4S * 16;
    info->device_bytes = s->device_bytes;
    info->builder = (uint32_t)s->builder;
    info->build_ms = s->build_ms;
    for (int k = 0; k < 3; k++) info->bounds_min[k] = s->host.root_lo[k], info->bounds_max[k] = s->host.root_hi[k];
    return RT_OK;
}

int rt_set_stage_timing(RtScene *s, int enabled) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    s->timing = enabled != 0;
    return RT_OK;
}

int rt_set_counting(RtScene *s, int enabled) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    s->counting = enabled != 0;
    return RT_OK;
}

// Reads the device-side counters (synchronises the device) and resets them.
int rt_scene_stats(const RtScene *cs, RtStats *out) {
    RtScene *s = const_cast<RtScene *>(cs);
    if (!s || !out) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    RT_HIP(hipSetDevice(s->device));
    RT_HIP(hipDeviceSynchronize());
    collect_timing(s);
    if (s->totals.p) {
        unsigned long long t[2] = {0, 0};
        RT_HIP(hipMemcpy(t, s->totals.p, sizeof(t), hipMemcpyDeviceToHost));
        s->stats.rays += t[0];
        s->stats.paths += t[1];
        RT_HIP(hipMemset(s->totals.p, 0, 64));
    }
    if (s->counters.p) {
        unsigned long long c[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        RT_HIP(hipMemcpy(c, s->counters.p, sizeof(c), hipMemcpyDeviceToHost));
        s->stats.node_visits += c[0];
        s->stats.prim_tests += c[1];
        s->stats.leaf_steps += c[2], s->stats.leaf_step_lanes += c[3];
        s->stats.node_steps += c[4], s->stats.node_step_lanes += c[5];
        s->stats.l1_lookups += c[14];
        if (std::getenv("RTCORE_DEBUG_STEPS") && c[2] + c[4])
            std::fprintf(stderr,
                         "[rtcore] leaf steps %llu with %.1f lanes, node steps %llu with %.1f lanes, %.1f lanes not stepping "
                         "per step, %llu refills\n",
                         c[2], c[2] ? (double)c[3] / (double)c[2] : 0.0, c[4], c[4] ? (double)c[5] / (double)c[4] : 0.0,
                         (double)c[6] / (double)(c[2] + c[4]), c[7]),
            std::fprintf(stderr, "[rtcore] refills take %.1f %% of the waves' time, %.0f clocks each\n",
                         100.0 * (double)c[8] / (double)(c[9] ? c[9] : 1), (double)c[8] / (double)(c[7] ? c[7] : 1));
        RT_HIP(hipMemset(s->counters.p, 0, 128));
    }
    *out = s->stats;
    s->stats = RtStats{};
    return RT_OK;
}

int rt_render_pass_device(RtScene *s, const RtRenderParams *p, float *d_accum, void *stream) {
    return render_device(s, p, d_accum, (hipStream_t)stream, false);
}
int rt_preview_device(RtScene *s, const RtRenderParams *p, float *d_out, void *stream) {
    return render_device(s, p, d_out, (hipStream_t)stream, true);
}

static int render_host(RtScene *s, const RtRenderParams *p, float *buf, bool preview) {
    int rc = check_params(s, p, buf);
    if (rc) return rc;
    RT_HIP(hipSetDevice(s->device));
    size_t bytes = (size_t)p->width * p->height * 3 * sizeof(float);
    if ((rc = s->accum_tmp.ensure(bytes))) return rc;
    // a pass adds to the caller's frame; a preview overwrites it, but only inside the pixel range it renders:
    // the rest of the caller's buffer must come back as it went in
    if (!preview || p->pixel_count != 0) RT_HIP(hipMemcpy(s->accum_tmp.p, buf, bytes, hipMemcpyHostToDevice));
    if ((rc = render_device(s, p, (float *)s->accum_tmp.p, nullptr, preview))) return rc;
    RT_HIP(hipMemcpy(buf, s->accum_tmp.p, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}
int rt_render_pass(RtScene *s, const RtRenderParams *p, float *accum) { return render_host(s, p, accum, false); }
int rt_preview(RtScene *s, const RtRenderParams *p, float *out) { return render_host(s, p, out, true); }

static int closest_hit_device(RtScene *s, const float *d_org_dir, float t_min, float t_max, size_t n, int32_t *d_id,
                              float *d_t, float *d_normal, float *d_point, hipStream_t stream, bool counted) {
    if (!s || (n && !d_org_dir)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(s->device));
    const size_t chunk_max = (size_t)default_paths_in_flight(s);  // large launches keep the tail of the persistent kernel small
    int rc = ensure_work(s, std::min(n, chunk_max));
    if (rc) return rc;
    for (size_t off = 0; off < n; off += chunk_max) {
        uint32_t m = (uint32_t)std::min(chunk_max, n - off);
        float4 *ro = (float4 *)s->st_o[0].p, *rd = (float4 *)s->st_d[0].p;
        {
            StageTimer tm(s, stream, 2);
            hipLaunchKernelGGL(pack_rays_kernel, dim3(blocks_for(m)), dim3(kBlock), 0, stream, d_org_dir + off * 6, m, ro, rd);
        }
        uint32_t *head = (uint32_t *)s->counts.p + kCountSlots + (size_t)(kCountSlots - 1) * kWorkWords;  // the last launch slot
        RT_HIP(hipMemsetAsync(head, 0, kWorkWords * sizeof(uint32_t), stream));
        if ((rc = launch_extend(s, stream, ro, rd, nullptr, m, t_min, t_max, counted, head))) return rc;
        {
            StageTimer tm(s, stream, 2);
            hipLaunchKernelGGL(unpack_hits_kernel, dim3(blocks_for(m)), dim3(kBlock), 0, stream,
                               s->view, (const float2 *)s->hit_a.p, ro, rd, m,
                               d_id ? d_id + off : nullptr, d_t ? d_t + off : nullptr,
                               d_normal ? d_normal + off * 3 : nullptr, d_point ? d_point + off * 3 : nullptr);
        }
        RT_HIP(hipGetLastError());
    }
    s->stats.rays += n;
    return RT_OK;
}

int rt_closest_hit_batch_device(RtScene *s, const float *d_org_dir, float t_min, float t_max, size_t n, int32_t *d_id,
                                float *d_t, float *d_normal, float *d_point, void *stream) {
    return closest_hit_device(s, d_org_dir, t_min, t_max, n, d_id, d_t, d_normal, d_point, (hipStream_t)stream, false);
}
int rt_closest_hit_batch_counted_device(RtScene *s, const float *d_org_dir, float t_min, float t_max, size_t n,
                                        int32_t *d_id, float *d_t, float *d_normal, float *d_point, void *stream) {
    return closest_hit_device(s, d_org_dir, t_min, t_max, n, d_id, d_t, d_normal, d_point, (hipStream_t)stream, true);
}

// The closest-hit batch for callers that cannot allocate: same kernels as closest_hit_device, work memory in the
// per-device `scratch` array.  Calls no hipMalloc / hipFree; no events, no counters on the scene.
int rt_closest_hit_batch_noalloc_device(RtScene *s, const float *d_org_dir, float t_min, float t_max, size_t n,
                                        int32_t *d_id, float *d_t, float *d_normal, float *d_point, void *stream_) {
    if (!s || (n && !d_org_dir)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (n > kNoAllocMaxRays)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_closest_hit_batch_noalloc_device takes at most " +
                                                 std::to_string(kNoAllocMaxRays) + " rays per call, got " + std::to_string(n));
    if (n == 0) return RT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    RT_HIP(hipSetDevice(s->device));
    unsigned char *base = nullptr;
    int rc = scratch_base(s->device, &base);
    if (rc) return rc;
    // the grid shrinks until the stacks' overflow area fits what is left of the array
    const uint32_t deep = s->view.max_stack > RT_LDS_STACK ? s->view.max_stack - RT_LDS_STACK : 0;
    uint32_t blocks = std::min(blocks_for(n), s->extend_blocks);
    if (deep) {
        const size_t per_block = (size_t)deep * kBlock * sizeof(uint2);
        const size_t fit = (kScratchBytes - kScratchOvf) / per_block;
        if (fit < 1)
            return fail(RT_ERR_INVALID_ARGUMENT, "the scene's tree is too deep for the no-allocation entry point (" +
                                                     std::to_string(s->view.max_stack) + " stack entries)");
        if (blocks > fit) blocks = (uint32_t)fit;
    }
    uint32_t *head = (uint32_t *)base;
    float4 *ro = (float4 *)(base + kScratchRo), *rd = (float4 *)(base + kScratchRd);
    float2 *hit = (float2 *)(base + kScratchHit);
    uint2 *ovf = (uint2 *)(base + kScratchOvf);
    const uint32_t m = (uint32_t)n;
    RT_HIP(hipMemsetAsync(head, 0, kScratchRo, stream));  // the array keeps the last call's contents
    hipLaunchKernelGGL(pack_rays_kernel, dim3(blocks_for(m)), dim3(kBlock), 0, stream, d_org_dir, m, ro, rd);
    auto kernel = s->compact ? extend_kernel<false, true> : extend_kernel<false, false>;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBlock), 0,
                       stream, s->view, (const float4 *)ro, (const float4 *)rd, (const uint32_t *)nullptr, m, t_min, t_max,
                       hit, ovf, head, (unsigned long long *)nullptr, s->compact ? kServiceLanesCompact : kServiceLanesWide);
    hipLaunchKernelGGL(unpack_hits_kernel, dim3(blocks_for(m)), dim3(kBlock), 0, stream, s->view, (const float2 *)hit,
                       (const float4 *)ro, (const float4 *)rd, m, d_id, d_t, d_normal, d_point);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

int rt_closest_hit_batch(RtScene *s, const float *org_dir, float t_min, float t_max, size_t n, int32_t *prim_id, float *t,
                         float *normal, float *point) {
    if (!s || (n && !org_dir)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(s->device));
    int rc;
    if ((rc = s->io_a.ensure(n * 24))) return rc;
    if ((rc = s->io_b.ensure(n * 4))) return rc;
    if ((rc = s->io_c.ensure(n * 4))) return rc;
    if ((rc = s->io_d.ensure(n * 12))) return rc;
    if ((rc = s->io_e.ensure(n * 12))) return rc;
    RT_HIP(hipMemcpy(s->io_a.p, org_dir, n * 24, hipMemcpyHostToDevice));
    if ((rc = closest_hit_device(s, (const float *)s->io_a.p, t_min, t_max, n, (int32_t *)s->io_b.p, (float *)s->io_c.p,
                                 (float *)s->io_d.p, (float *)s->io_e.p, nullptr, false)))
        return rc;
    RT_HIP(hipDeviceSynchronize());
    if (prim_id) RT_HIP(hipMemcpy(prim_id, s->io_b.p, n * 4, hipMemcpyDeviceToHost));
    if (t) RT_HIP(hipMemcpy(t, s->io_c.p, n * 4, hipMemcpyDeviceToHost));
    if (normal) RT_HIP(hipMemcpy(normal, s->io_d.p, n * 12, hipMemcpyDeviceToHost));
    if (point) RT_HIP(hipMemcpy(point, s->io_e.p, n * 12, hipMemcpyDeviceToHost));
    return RT_OK;
}

// to_rgb / aces_tonemap (src/render.rs:19-55); host arithmetic only
uint32_t rt_to_rgb(const float rgb[3], float inv_gamma) {
    const float m1[3][3] = {{0.59719f, 0.07600f, 0.02840f}, {0.35458f, 0.90834f, 0.13383f}, {0.04823f, 0.01566f, 0.83777f}};
    const float m2[3][3] = {{1.60475f, -0.10208f, -0.00327f}, {-0.53108f, 1.10813f, -0.07276f}, {-0.07367f, -0.00605f, 1.07602f}};
    float v[3], q[3], o[3];
    for (int k = 0; k < 3; k++) v[k] = m1[0][k] * rgb[0] + m1[1][k] * rgb[1] + m1[2][k] * rgb[2];
    for (int k = 0; k < 3; k++) {
        float a = v[k] * (v[k] + 0.0245786f) - 0.000090537f;
        float b = v[k] * (0.983729f * v[k] + 0.432951f) + 0.238081f;
        q[k] = a / b;
    }
    uint32_t out = 255u << 24;
    for (int k = 0; k < 3; k++) {
        o[k] = m2[0][k] * q[0] + m2[1][k] * q[1] + m2[2][k] * q[2];
        o[k] = std::fmin(std::fmax(o[k], 0.0f), 1.0f);
        o[k] = (float)std::pow((double)o[k], (double)inv_gamma);  // as the device kernel does (see post.hip)
        out |= ((uint32_t)(o[k] * 255.4f)) << (16 - 8 * k);
    }
    return out;
}

}  // extern "C"
