// rt_pool.hpp -- closest hit with a pool of rays per wave (the second form of the extend stage).
//
// The first form (extend_kernel) binds one ray to one lane; a wave then runs a node step for the
// lanes whose ray is at a node or a leaf step for those at a leaf, and about half its lanes sit
// out every step.  Here a wave owns a pool of kPoolSlots rays whose whole state lives in LDS
// (origin, 1/direction, direction, best hit, current entry, stack), and every step takes up to 64
// rays that all want the SAME kind of step from a queue of slot numbers -- lanes are workers, not
// owners.  Steps are software-pipelined one deep: the records of the next bundle (tree nodes,
// primitives or new rays) are requested before the current bundle is worked on, so a wave covers
// its own memory latency and two waves per SIMD suffice (the pools fill the 160 KB of LDS).
//
// Per ray the arithmetic is that of rt_traverse.hpp (same functions), so hit records are the same
// bits; only which lane does the work, and when, differs.
#ifndef RT_POOL_HPP
#define RT_POOL_HPP

#include "rt_traverse.hpp"

#ifndef RT_POOL_SLOTS
#define RT_POOL_SLOTS 160
#endif
#ifndef RT_POOL_STACK
#define RT_POOL_STACK 8
#endif
#ifndef RT_POOL_POP_TURNS
#define RT_POOL_POP_TURNS 2
#endif
#ifndef RT_POOL_PARTIAL
#define RT_POOL_PARTIAL 48
#endif

namespace rtd {

constexpr uint32_t kPoolSlots = RT_POOL_SLOTS;  // at most 256: queue entries are bytes
constexpr uint32_t kPoolStack = RT_POOL_STACK;  // stack entries per ray kept in LDS; deeper ones go to HBM
static_assert(kPoolSlots <= 256 && kPoolSlots >= 128, "pool size");

// One wave's pool.  A/B/C are read 16 bytes at a time; single words of them are rewritten in place.
struct __align__(16) PoolLds {
    float4 A[kPoolSlots];                 // origin, best t
    float4 B[kPoolSlots];                 // 1 / direction, current entry (child word, bits)
    float4 C[kPoolSlots];                 // direction, best index (visit rank, bits)
    uint2 D[kPoolSlots];                  // x: stack entries | best.ties << 31, y: ray index
    uint2 stack[kPoolStack][kPoolSlots];  // (child word, entry distance), entry-major
    uint8_t q[4][256];                    // ring buffers of slot numbers: node, leaf, pop, free
};

enum { Q_NODE = 0, Q_LEAF = 1, Q_POP = 2, Q_FREE = 3, K_REFILL = 3, K_NONE = 4 };

struct PoolQueues {  // wave-uniform
    uint32_t tail[4], count[4];
};

// Every lane with dest < 4 appends its slot number to queue `dest` (one byte store per lane for all four queues).
RT_DEV void pool_route(PoolLds &L, PoolQueues &Q, uint32_t dest, uint32_t slot) {
    const unsigned long long m0 = __ballot(dest == 0u), m1 = __ballot(dest == 1u), m2 = __ballot(dest == 2u),
                             m3 = __ballot(dest == 3u);
    const unsigned long long mine = dest == 0u ? m0 : (dest == 1u ? m1 : (dest == 2u ? m2 : m3));
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
    const uint32_t t0 = Q.tail[0], t1 = Q.tail[1], t2 = Q.tail[2], t3 = Q.tail[3];
    uint32_t tail = t3;
    tail = dest == 2u ? t2 : tail;
    tail = dest == 1u ? t1 : tail;
    tail = dest == 0u ? t0 : tail;
    if (dest < 4u) (&L.q[0][0])[dest * 256u + ((tail + rank) & 255u)] = (uint8_t)slot;
    const uint32_t c0 = (uint32_t)__popcll(m0), c1 = (uint32_t)__popcll(m1), c2 = (uint32_t)__popcll(m2), c3 = (uint32_t)__popcll(m3);
    Q.tail[0] += c0, Q.count[0] += c0;
    Q.tail[1] += c1, Q.count[1] += c1;
    Q.tail[2] += c2, Q.count[2] += c2;
    Q.tail[3] += c3, Q.count[3] += c3;
}
constexpr uint32_t kNoDest = 4u;
// the n oldest entries of queue `k`, one per lane (lanes >= n get nothing)
RT_DEV uint32_t pool_take(PoolLds &L, PoolQueues &Q, int k, uint32_t n, uint32_t lane) {
    const uint32_t head = Q.tail[k] - Q.count[k];
    uint32_t slot = 0;
    if (lane < n) slot = L.q[k][(head + lane) & 255u];
    Q.count[k] -= n;
    return slot;
}

typedef float V4 __attribute__((ext_vector_type(4)));
RT_DEV float4 f4(V4 v) { return make_float4(v.x, v.y, v.z, v.w); }

struct PoolBundle {
    uint32_t kind;  // wave-uniform
    uint32_t n;     // wave-uniform: lanes [0, n) hold a ray
    uint32_t slot;
    uint32_t entry;  // node / leaf bundles: the ray's current child word; refill: the ray index
    // The bundle's records.  They are requested by an asm statement (pool_request) that the compiler does not count
    // among the outstanding memory operations, and become valid at pool_await: nothing may read them in between.
    V4 r0, r1, r2, r3, r4, r5, r6;
};

// Seven 16-byte loads per lane: words 0, 2..6 of the record at P and word 0 of the record at P1.  Written in
// assembly so that the compiler's own wait insertion leaves them in flight: with the loads of two bundles and its
// own occasional loads and stores in one loop it falls back to waiting for everything at every use.
RT_DEV void pool_request(PoolBundle &b, const float4 *P, const float4 *P1) {
    asm volatile(
        "global_load_dwordx4 %0, %7, off\n\t"
        "global_load_dwordx4 %1, %8, off\n\t"
        "global_load_dwordx4 %2, %7, off offset:32\n\t"
        "global_load_dwordx4 %3, %7, off offset:48\n\t"
        "global_load_dwordx4 %4, %7, off offset:64\n\t"
        "global_load_dwordx4 %5, %7, off offset:80\n\t"
        "global_load_dwordx4 %6, %7, off offset:96"
        : "=&v"(b.r0), "=&v"(b.r1), "=&v"(b.r2), "=&v"(b.r3), "=&v"(b.r4), "=&v"(b.r5), "=&v"(b.r6)
        : "v"(P), "v"(P1)
        : "memory");
}
// Waits until the records of `b` have arrived.  The wave issues memory operations in order and the NEXT bundle's
// seven loads are always issued between pool_request(b) and pool_await(b), so "at most seven operations still
// outstanding" means that b's are complete, whatever else (result stores, stack overflow traffic) was issued too.
RT_DEV void pool_await(PoolBundle &b) {
    asm volatile("s_waitcnt vmcnt(7)"
                 : "+v"(b.r0), "+v"(b.r1), "+v"(b.r2), "+v"(b.r3), "+v"(b.r4), "+v"(b.r5), "+v"(b.r6)
                 :
                 : "memory");
}

struct PoolCounters {
    uint32_t visits, tests;
    unsigned long long steps[4], lanes[4];
};

template <bool COUNT>
struct PoolWave {
    PoolLds &L;
    PoolQueues Q;
    const RtSceneView &sc;
    const float4 *__restrict__ ro;
    const float4 *__restrict__ rd;
    float2 *__restrict__ hit;
    uint2 *ovf;            // this wave's part of the overflow area: [entry - kPoolStack][kPoolSlots]
    uint32_t *head;
    uint32_t n_rays;
    float t_min, t_max;
    uint32_t lane;
    uint32_t chunk_next, chunk_end;
    bool no_more;
    PoolCounters cnt;

    RT_DEV PoolWave(PoolLds &l, const RtSceneView &s) : L(l), sc(s) {}

    // The overflow area is addressed as global memory explicitly: through generic pointers the compiler folds
    // "LDS or overflow" into one flat access, and a flat access waits for every outstanding memory operation.
    typedef __attribute__((address_space(1))) unsigned long long GlobalU64;
    RT_DEV void stack_store(uint32_t slot, uint32_t e, uint32_t w, float t) {
        if (e < kPoolStack)
            L.stack[e][slot] = make_uint2(w, __float_as_uint(t));
        else
            ((GlobalU64 *)ovf)[(size_t)(e - kPoolStack) * kPoolSlots + slot] =
                (unsigned long long)w | ((unsigned long long)__float_as_uint(t) << 32);
    }
    RT_DEV uint2 stack_load(uint32_t slot, uint32_t e) {
        if (e < kPoolStack) return L.stack[e][slot];
        const unsigned long long v = ((const GlobalU64 *)ovf)[(size_t)(e - kPoolStack) * kPoolSlots + slot];
        return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    }

    // Chooses the next bundle and requests its records.  Every call issues the same seven 16-byte loads on every
    // lane, whatever the kind (spare ones re-read a record that is in cache anyway): only then can the wave wait for
    // the PREVIOUS bundle's records by count and leave these in flight while it works on them.
    RT_DEV void select(PoolBundle &b) {
        uint32_t kind = K_NONE;
        const bool can_refill = !no_more && Q.count[Q_FREE] > 0;
        if (Q.count[Q_LEAF] >= 64u)
            kind = Q_LEAF;
        else if (Q.count[Q_NODE] >= 64u)
            kind = Q_NODE;
        else if (can_refill && Q.count[Q_FREE] >= RT_POOL_PARTIAL)
            kind = K_REFILL;
        else if (Q.count[Q_POP] >= RT_POOL_PARTIAL)
            kind = Q_POP;
        else {
            // nothing fills a wave: the longest queue
            uint32_t best = 0;
            if (Q.count[Q_NODE] > best) best = Q.count[Q_NODE], kind = Q_NODE;
            if (Q.count[Q_LEAF] > best) best = Q.count[Q_LEAF], kind = Q_LEAF;
            if (Q.count[Q_POP] > best) best = Q.count[Q_POP], kind = Q_POP;
            if (can_refill && Q.count[Q_FREE] > best) best = Q.count[Q_FREE], kind = K_REFILL;
        }
        if (kind == K_REFILL && chunk_next == chunk_end) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(head, 256u);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            chunk_next = base < n_rays ? base : n_rays;
            chunk_end = (base + 256u) < n_rays ? (base + 256u) : n_rays;
            if (base >= n_rays) {
                no_more = true;
                kind = K_NONE;
            }
        }
        const float4 *P = (const float4 *)sc.nodes;  // a record every lane may read: node 0
        const float4 *P1 = P + 1;
        uint32_t n = 0, slot = 0, entry = 0;
        // (constant queue numbers: an index known only at run time would put the counters in scratch memory)
        if (kind == K_REFILL) {
            n = Q.count[Q_FREE] < 64u ? Q.count[Q_FREE] : 64u;
            const uint32_t avail = chunk_end - chunk_next;
            if (n > avail) n = avail;
            slot = pool_take(L, Q, Q_FREE, n, lane);
            entry = chunk_next + lane;
            chunk_next += n;
            if (lane < n) P = ro + entry, P1 = rd + entry;
        } else if (kind == Q_NODE) {
            n = Q.count[Q_NODE] < 64u ? Q.count[Q_NODE] : 64u;
            slot = pool_take(L, Q, Q_NODE, n, lane);
            if (lane < n) {
                entry = __float_as_uint(L.B[slot].w);
                P = (const float4 *)sc.nodes + (size_t)entry * RT_NODE_FLOAT4S;
                P1 = P + 1;
            }
        } else if (kind == Q_LEAF) {
            n = Q.count[Q_LEAF] < 64u ? Q.count[Q_LEAF] : 64u;
            slot = pool_take(L, Q, Q_LEAF, n, lane);
            if (lane < n) {
                entry = __float_as_uint(L.B[slot].w);
                P = (const float4 *)sc.prims + (size_t)(entry & RT_LEAF_FIRST_MASK) * RT_PRIM_FLOAT4S;
                P1 = P + 1;
            }
        } else if (kind == Q_POP) {
            n = Q.count[Q_POP] < 64u ? Q.count[Q_POP] : 64u;
            slot = pool_take(L, Q, Q_POP, n, lane);
        }
        b.kind = kind, b.n = n, b.slot = slot, b.entry = entry;
        // node: its seven words; leaf: two primitive records and a spare word (the arrays are padded for it);
        // new rays: origin and direction, the spare words are neighbours' origins
        pool_request(b, P, P1);
    }

    template <bool DEEP>
    RT_DEV void pop_turns(uint32_t slot, float best_t, uint32_t &sp, bool &popping, bool &finished, uint32_t &entry) {
#pragma unroll
        for (int turn = 0; turn < RT_POOL_POP_TURNS; turn++) {
            if (popping) {
                if (sp == 0u) {
                    finished = true;
                    popping = false;
                } else {
                    const uint32_t below = sp >= 2u ? sp - 2u : 0u;
                    uint2 e1, e2;
                    if (!DEEP)
                        e1 = L.stack[sp - 1u][slot], e2 = L.stack[below][slot];
                    else
                        e1 = stack_load(slot, sp - 1u), e2 = stack_load(slot, below);
                    if (__uint_as_float(e1.y) <= best_t) {
                        entry = e1.x;
                        sp -= 1u;
                        popping = false;
                    } else if (sp >= 2u && __uint_as_float(e2.y) <= best_t) {
                        entry = e2.x;
                        sp -= 2u;
                        popping = false;
                    } else {
                        sp = below;
                    }
                }
            }
        }
    }

    // The rays with `want` take their next stack entry: up to RT_POOL_POP_TURNS turns of two entries each.  Returns
    // the queue the ray goes to next: node, leaf, still popping, or free (finished: result written).
    RT_DEV uint32_t pop_next(bool want, uint32_t slot, float best_t, int32_t best_index, uint32_t sp, uint32_t ties_bit) {
        if (__ballot(want) == 0ull) return kNoDest;
        bool popping = want, finished = false;
        uint32_t entry = 0;
        // Two copies of the turns: one that only ever reads LDS, for waves none of whose rays has entries in the
        // overflow area (the usual case), and the general one.  A load from the overflow area, even on a path that
        // no lane takes, puts full memory waits on the common path, and those would drain the requested records.
        if (__ballot(want && sp > kPoolStack) == 0ull)
            pop_turns<false>(slot, best_t, sp, popping, finished, entry);
        else
            pop_turns<true>(slot, best_t, sp, popping, finished, entry);
        uint32_t dest = kNoDest;
        if (want) {
            L.D[slot].x = sp | ties_bit;
            if (finished) {
                hit[L.D[slot].y] = make_float2(best_t, __int_as_float(best_index));
                dest = Q_FREE;
            } else if (popping) {
                dest = Q_POP;
            } else {
                L.B[slot].w = __uint_as_float(entry);
                dest = (entry & RT_CHILD_LEAF) != 0u ? Q_LEAF : Q_NODE;
            }
        }
        return dest;
    }

    // Works on a bundle whose records were requested by select().
    RT_DEV void consume(PoolBundle &b) {
        pool_await(b);
        if (b.kind == K_NONE) return;
        const bool valid = lane < b.n;
        const uint32_t slot = b.slot;
        if (COUNT && lane == 0) {
            if (b.kind == Q_NODE) cnt.steps[Q_NODE]++, cnt.lanes[Q_NODE] += b.n;
            if (b.kind == Q_LEAF) cnt.steps[Q_LEAF]++, cnt.lanes[Q_LEAF] += b.n;
            if (b.kind == Q_POP) cnt.steps[Q_POP]++, cnt.lanes[Q_POP] += b.n;
            if (b.kind == K_REFILL) cnt.steps[K_REFILL]++, cnt.lanes[K_REFILL] += b.n;
        }
        if (b.kind == Q_NODE) {
            bool want_pop = false;
            float best_t = 0.0f;
            int32_t best_index = -1;
            uint32_t sp = 0, ties_bit = 0, dest = kNoDest;
            if (valid) {
                const float4 a = L.A[slot], bb = L.B[slot];
                const uint32_t d0 = L.D[slot].x;
                best_t = a.w;
                sp = d0 & 0x7fffffffu, ties_bit = d0 & 0x80000000u;
                float key[4];
                uint32_t cw[4];
                if (COUNT) cnt.visits++;
                const int nh = node_children(f4(b.r0), f4(b.r1), f4(b.r2), f4(b.r3), f4(b.r4), f4(b.r5), f4(b.r6), f3(a.x, a.y, a.z), f3(bb.x, bb.y, bb.z),
                                             t_min, best_t, key, cw);
                if (nh > 1) {
                    const uint32_t top = sp + (uint32_t)nh - 1u;
                    if (top <= kPoolStack) {
                        if (nh > 1) L.stack[top - 1u][slot] = make_uint2(cw[1], __float_as_uint(key[1]));
                        if (nh > 2) L.stack[top - 2u][slot] = make_uint2(cw[2], __float_as_uint(key[2]));
                        if (nh > 3) L.stack[top - 3u][slot] = make_uint2(cw[3], __float_as_uint(key[3]));
                    } else {
                        if (nh > 1) stack_store(slot, top - 1u, cw[1], key[1]);
                        if (nh > 2) stack_store(slot, top - 2u, cw[2], key[2]);
                        if (nh > 3) stack_store(slot, top - 3u, cw[3], key[3]);
                    }
                    sp = top;
                    L.D[slot].x = sp | ties_bit;
                }
                want_pop = nh == 0;
                if (nh > 0) {
                    L.B[slot].w = __uint_as_float(cw[0]);
                    dest = (cw[0] & RT_CHILD_LEAF) != 0u ? Q_LEAF : Q_NODE;
                } else {
                    best_index = __float_as_int(L.C[slot].w);
                }
            }
            const uint32_t pdest = pop_next(valid && want_pop, slot, best_t, best_index, sp, ties_bit);
            pool_route(L, Q, want_pop ? pdest : dest, slot);
        } else if (b.kind == Q_LEAF) {
            float best_t = 0.0f;
            int32_t best_index = -1;
            uint32_t sp = 0, ties_bit = 0;
            if (valid) {
                const float4 a = L.A[slot], bb = L.B[slot], c = L.C[slot];
                const uint32_t d0 = L.D[slot].x;
                sp = d0 & 0x7fffffffu;
                Ray ray;
                ray.o = f3(a.x, a.y, a.z);
                ray.d = f3(c.x, c.y, c.z);
                const F3 inv = f3(bb.x, bb.y, bb.z);
                Best best{a.w, __float_as_int(c.w), (d0 >> 31) != 0u};
                const int32_t first = (int32_t)(b.entry & RT_LEAF_FIRST_MASK);
                const int32_t count = (int32_t)((b.entry >> 28) & 7u) + 1;
                if (COUNT) cnt.tests += (uint32_t)count;
                // a leaf holds one or two primitives (the builder refuses more)
                test_record(f4(b.r0), f4(b.r1), f4(b.r2), first, ray, inv, t_min, t_max, best);
                if (count > 1) test_record(f4(b.r3), f4(b.r4), f4(b.r5), first + 1, ray, inv, t_min, t_max, best);
                if (best.index != __float_as_int(c.w) || best.t != a.w) {
                    L.A[slot].w = best.t;
                    L.C[slot].w = __int_as_float(best.index);
                }
                best_t = best.t, best_index = best.index;
                ties_bit = best.ties ? 0x80000000u : 0u;
            }
            pool_route(L, Q, pop_next(valid, slot, best_t, best_index, sp, ties_bit), slot);
        } else if (b.kind == Q_POP) {
            float best_t = 0.0f;
            int32_t best_index = -1;
            uint32_t sp = 0, ties_bit = 0;
            if (valid) {
                const uint32_t d0 = L.D[slot].x;
                sp = d0 & 0x7fffffffu, ties_bit = d0 & 0x80000000u;
                best_t = L.A[slot].w;
                best_index = __float_as_int(L.C[slot].w);
            }
            pool_route(L, Q, pop_next(valid, slot, best_t, best_index, sp, ties_bit), slot);
        } else {  // K_REFILL: Ray::new has been applied by the producer of ro / rd; Bvh::hit's first box test (src/tracer/bvh.rs:144-147)
            bool start = false;
            if (valid) {
                Traversal tr;
                tr.ray.o = f3(b.r0.x, b.r0.y, b.r0.z);
                tr.ray.d = f3(b.r1.x, b.r1.y, b.r1.z);
                Stack unused;
                unused.sp = 0;
                start = traversal_begin<COUNT>(sc, tr, unused, t_min, t_max, &cnt.visits);
                if (start) {
                    L.A[slot] = make_float4(tr.ray.o.x, tr.ray.o.y, tr.ray.o.z, t_max);
                    L.B[slot] = make_float4(tr.inv.x, tr.inv.y, tr.inv.z, __uint_as_float(0u));  // entry: node 0
                    L.C[slot] = make_float4(tr.ray.d.x, tr.ray.d.y, tr.ray.d.z, __int_as_float(-1));
                    L.D[slot] = make_uint2(0u, b.entry);
                } else {
                    hit[b.entry] = make_float2(0.0f, __int_as_float(-1));
                }
            }
            pool_route(L, Q, valid ? (start ? (uint32_t)Q_NODE : (uint32_t)Q_FREE) : kNoDest, slot);
        }
    }

    RT_DEV void run() {
        PoolBundle a, b;
        a.kind = K_NONE, a.n = 0, a.slot = 0, a.entry = 0;
        pool_request(a, (const float4 *)sc.nodes, (const float4 *)sc.nodes);  // every consume() awaits a request
        for (;;) {
            select(b);
            consume(a);
            if (b.kind == K_NONE && a.kind == K_NONE && queues_empty()) break;
            select(a);
            consume(b);
            if (b.kind == K_NONE && a.kind == K_NONE && queues_empty()) break;
        }
    }
    // nothing queued and no ray left to start (the free queue holds every slot then)
    RT_DEV bool queues_empty() const {
        return Q.count[Q_NODE] == 0u && Q.count[Q_LEAF] == 0u && Q.count[Q_POP] == 0u && (no_more || Q.count[Q_FREE] == 0u);
    }
};

}  // namespace rtd
#endif
