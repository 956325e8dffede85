// sim.cpp -- prices an 8-wide form of the tree on the CPU before anyone builds it for the GPU.
//
// Takes a scene description through the same host build as the library (csrc/scene_build.cpp), shoots rays that
// look like a path tracer's secondary rays (cosine-weighted directions from random points of random triangles) and
// counts, per ray, what the closest-hit kernel pays for by the visit:
//   * the 4-wide tree as the kernel walks it: children sorted by entry distance, far hits pushed with their entry
//     distance, entries beyond the best hit dropped at pop;
//   * the same tree collapsed to 8-wide nodes (a node's inner child with the largest box is replaced by its children
//     while the result fits eight slots), walked (a) with the hit children sorted by entry distance, (b) in an order
//     fixed per ray octant (children ordered by their box centre along the octant's diagonal: the best an
//     octant-ordered node can do), both with the entry distances kept for the drop at pop.
// Exact child boxes throughout (the compact nodes' outward rounding adds ~1.6 % node visits to either form).
// Build: see tools/sim_wide/run.py.  Not part of the product or its tests.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rtcore.h"
#include "scene_build.hpp"

namespace {

struct V3 {
    float x, y, z;
};
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 norm(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

struct Rng {
    uint64_t s;
    uint32_t next() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (uint32_t)(s >> 33);
    }
    float uni() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
};

struct Box {
    float lo[3], hi[3];
};
inline bool slab(const Box &b, V3 o, V3 inv, float t_min, float t_best, float *tn_out) {
    float t0x = (b.lo[0] - o.x) * inv.x, t1x = (b.hi[0] - o.x) * inv.x;
    float t0y = (b.lo[1] - o.y) * inv.y, t1y = (b.hi[1] - o.y) * inv.y;
    float t0z = (b.lo[2] - o.z) * inv.z, t1z = (b.hi[2] - o.z) * inv.z;
    float tn = std::fmax(std::fmax(std::fmax(std::fmin(t0x, t1x), std::fmin(t0y, t1y)), std::fmin(t0z, t1z)), t_min);
    float tf = std::fmin(std::fmin(std::fmin(std::fmax(t0x, t1x), std::fmax(t0y, t1y)), std::fmax(t0z, t1z)), t_best);
    *tn_out = tn;
    return tn <= tf;
}

// a node of either width: child boxes, child words (as in the 4-wide form: node index or leaf word), and for the
// octant order the permutation of the children per ray octant
struct WNode {
    int n = 0;
    Box box[8];
    uint32_t word[8];
    uint8_t order[8][8];  // order[octant][k] = k-th child to visit (nearest first) for rays of that octant
};

struct Tree {
    std::vector<WNode> nodes;
};

struct Counts {
    double node_visits = 0, leaf_visits = 0, prim_tests = 0, pushes = 0, max_sp = 0, sp_sum = 0, box_tests = 0, hits = 0;
    double sp_over[16] = {0};
    void add(const Counts &c) {
        node_visits += c.node_visits, leaf_visits += c.leaf_visits, prim_tests += c.prim_tests, pushes += c.pushes;
        max_sp = std::max(max_sp, c.max_sp), sp_sum += c.sp_sum, box_tests += c.box_tests, hits += c.hits;
        for (int k = 0; k < 16; k++) sp_over[k] += c.sp_over[k];
    }
};

struct Scene {
    const rtcore::HostScene *hs;
    const uint32_t *prims;  // 12 words per primitive
};

inline bool tri_hit(const uint32_t *rec, V3 o, V3 d, float t_min, float t_max, float *t_out) {
    float f[12];
    std::memcpy(f, rec, 48);
    V3 v0{f[0], f[1], f[2]}, e1{f[3], f[4], f[5]}, e2{f[6], f[7], f[8]};
    const bool two_sided = (rec[10] & RT_PRIM_FLAG) != 0;
    V3 h = cross(d, e2);
    float a = dot(e1, h);
    if ((!two_sided && a < 0) || std::fabs(a) < 1e-6f) return false;
    float inv = 1.0f / a;
    V3 s = o - v0;
    float u = inv * dot(s, h);
    if (u < 0 || u > 1) return false;
    V3 q = cross(s, e1);
    float v = inv * dot(d, q);
    if (v < 0 || u + v > 1) return false;
    float t = inv * dot(e2, q);
    if (t < t_min || t > t_max) return false;
    *t_out = t;
    return true;
}

// mode 0: hit children sorted by entry distance; mode 1: fixed order per octant
Counts trace(const Tree &T, const Scene &sc, V3 o, V3 d, int mode) {
    Counts c;
    const float t_min = 1.5e-4f;
    float best = 1e5f;
    V3 inv{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    const int oct = (d.x < 0 ? 1 : 0) | (d.y < 0 ? 2 : 0) | (d.z < 0 ? 4 : 0);
    struct E {
        uint32_t w;
        float t;
    };
    E stack[256];
    int sp = 0;
    uint32_t cur = 0;  // node 0
    bool have = true;
    uint64_t steps = 0;
    while (have) {
        if (cur & RT_CHILD_LEAF) {
            c.leaf_visits++;
            const uint32_t first = cur & RT_LEAF_FIRST_MASK, count = ((cur >> 28) & 7u) + 1;
            for (uint32_t k = 0; k < count; k++) {
                const uint32_t *rec = sc.prims + (size_t)(first + k) * 12;
                c.prim_tests++;
                if ((rec[10] & RT_PRIM_TYPE_MASK) != RT_PRIM_TRIANGLE) continue;  // (the room's boxes: ignored, the meshes are what counts)
                float t;
                if (tri_hit(rec, o, d, t_min, best, &t)) best = t, c.hits = 1;
            }
        } else {
            c.node_visits++;
            const WNode &n = T.nodes[cur];
            float key[8];
            uint32_t w[8];
            int nh = 0;
            if (mode == 0) {
                for (int k = 0; k < n.n; k++) {
                    float tn;
                    c.box_tests++;
                    if (slab(n.box[k], o, inv, t_min, best, &tn)) key[nh] = tn, w[nh] = n.word[k], nh++;
                }
                // nearest first
                for (int i = 1; i < nh; i++)
                    for (int j = i; j > 0 && key[j] < key[j - 1]; j--) std::swap(key[j], key[j - 1]), std::swap(w[j], w[j - 1]);
            } else {
                for (int k = 0; k < n.n; k++) {
                    const int ch = n.order[oct][k];
                    float tn;
                    c.box_tests++;
                    if (slab(n.box[ch], o, inv, t_min, best, &tn)) key[nh] = tn, w[nh] = n.word[ch], nh++;
                }
            }
            // far ones pushed farthest first, nearest continues
            for (int k = nh - 1; k >= 1; k--) stack[sp++] = E{w[k], key[k]}, c.pushes++;
            if (sp > c.max_sp) c.max_sp = sp;
            if (nh > 0) {
                cur = w[0];
                c.sp_sum += sp, steps++;
                for (int k = 0; k < 16; k++)
                    if (sp > k) c.sp_over[k]++;
                continue;
            }
        }
        // pop
        have = false;
        while (sp > 0) {
            E e = stack[--sp];
            if (e.t <= best) {
                cur = e.w, have = true;
                break;
            }
        }
    }
    return c;
}

Tree tree4(const rtcore::HostScene &hs) {
    Tree T;
    T.nodes.resize(hs.n_nodes);
    for (size_t i = 0; i < hs.n_nodes; i++) {
        const float *f = &hs.nodes[i * 32];
        uint32_t w[4];
        std::memcpy(w, f + 24, 16);
        WNode &n = T.nodes[i];
        for (int c = 0; c < 4; c++) {
            if (w[c] == RT_CHILD_NONE) continue;
            Box b;
            for (int a = 0; a < 3; a++) b.lo[a] = f[a * 4 + c], b.hi[a] = f[12 + a * 4 + c];
            n.box[n.n] = b, n.word[n.n] = w[c], n.n++;
        }
    }
    return T;
}

inline float half_area(const Box &b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

void set_orders(WNode &n) {
    for (int oct = 0; oct < 8; oct++) {
        float key[8];
        for (int k = 0; k < n.n; k++) {
            float cx = n.box[k].lo[0] + n.box[k].hi[0], cy = n.box[k].lo[1] + n.box[k].hi[1], cz = n.box[k].lo[2] + n.box[k].hi[2];
            key[k] = ((oct & 1) ? -cx : cx) + ((oct & 2) ? -cy : cy) + ((oct & 4) ? -cz : cz);
            n.order[oct][k] = (uint8_t)k;
        }
        std::sort(n.order[oct], n.order[oct] + n.n, [&](uint8_t a, uint8_t b) { return key[a] < key[b]; });
    }
}

// 8-wide collapse of the 4-wide tree: returns the new tree (node 0 = root)
Tree collapse8(const Tree &T4, int width) {
    Tree T;
    std::vector<uint32_t> todo;  // 4-wide node indices that root a wide node, in the order of their new indices
    std::vector<uint32_t> new_index(T4.nodes.size(), 0xFFFFFFFFu);
    todo.push_back(0);
    new_index[0] = 0;
    for (size_t q = 0; q < todo.size(); q++) {
        WNode n = T4.nodes[todo[q]];  // start from the four children
        for (;;) {
            int pick = -1;
            float best_area = -1;
            for (int k = 0; k < n.n; k++) {
                if (n.word[k] & RT_CHILD_LEAF) continue;
                const WNode &ch = T4.nodes[n.word[k]];
                if (n.n - 1 + ch.n > width) continue;
                float a = half_area(n.box[k]);
                if (a > best_area) best_area = a, pick = k;
            }
            if (pick < 0) break;
            const WNode ch = T4.nodes[n.word[pick]];
            n.box[pick] = n.box[n.n - 1], n.word[pick] = n.word[n.n - 1], n.n--;
            for (int k = 0; k < ch.n; k++) n.box[n.n] = ch.box[k], n.word[n.n] = ch.word[k], n.n++;
        }
        for (int k = 0; k < n.n; k++)
            if (!(n.word[k] & RT_CHILD_LEAF)) {
                const uint32_t old = n.word[k];
                new_index[old] = (uint32_t)todo.size();
                todo.push_back(old);
                n.word[k] = new_index[old];
            }
        set_orders(n);
        T.nodes.push_back(n);
    }
    return T;
}

}  // namespace

// out[0..]: per form (4-wide sorted, W-wide sorted, W-wide octant order, 4-wide octant order) 8 numbers each:
// node visits, leaf visits, primitive tests, pushes, mean stack depth at a node step, max stack depth, box tests, hit rate
// then per form 16 numbers: share of node steps with more than k entries on the stack, k = 0..15
extern "C" int sim_run(const RtSceneDesc *desc, uint64_t n_rays, uint32_t seed, int width, double *out, double *out_nodes) {
    rtcore::HostScene hs;
    std::string err;
    int rc = rtcore::build_host_scene(desc, hs, err);
    if (rc != RT_OK) {
        std::fprintf(stderr, "build failed: %s\n", err.c_str());
        return rc;
    }
    Scene sc{&hs, hs.prims.data()};
    Tree T4 = tree4(hs);
    for (auto &n : T4.nodes) set_orders(n);
    Tree T8 = collapse8(T4, width);
    out_nodes[0] = (double)T4.nodes.size(), out_nodes[1] = (double)T8.nodes.size();
    double fill = 0;
    for (auto &n : T8.nodes) fill += n.n;
    out_nodes[2] = fill / (double)T8.nodes.size();
    const int n_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::vector<Counts>> per(n_threads, std::vector<Counts>(4));
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&, t]() {
            Rng rng{(uint64_t)seed * 1000003ull + (uint64_t)t * 7919ull + 1ull};
            for (uint64_t r = (uint64_t)t; r < n_rays; r += (uint64_t)n_threads) {
                // a random triangle, a random point of it, a cosine-weighted direction about its geometric normal
                const uint32_t *rec;
                do {
                    rec = sc.prims + (size_t)(rng.next() % hs.n_prims) * 12;
                } while ((rec[10] & RT_PRIM_TYPE_MASK) != RT_PRIM_TRIANGLE);
                float f[12];
                std::memcpy(f, rec, 48);
                V3 v0{f[0], f[1], f[2]}, e1{f[3], f[4], f[5]}, e2{f[6], f[7], f[8]};
                float u = rng.uni(), v = rng.uni();
                if (u + v > 1) u = 1 - u, v = 1 - v;
                V3 o = v0 + e1 * u + e2 * v;
                V3 nn = cross(e1, e2);
                if (dot(nn, nn) == 0) continue;
                nn = norm(nn);
                V3 a = std::fabs(nn.x) > 0.9f ? V3{0, 1, 0} : V3{1, 0, 0};
                V3 tv = norm(cross(nn, a)), tu = cross(nn, tv);
                float r1 = rng.uni(), r2 = rng.uni();
                float phi = 6.2831853f * r1, sr = std::sqrt(r2), z = std::sqrt(1 - r2);
                V3 d = norm(tu * (std::cos(phi) * sr) + tv * (std::sin(phi) * sr) + nn * z);
                per[t][0].add(trace(T4, sc, o, d, 0));
                per[t][1].add(trace(T8, sc, o, d, 0));
                per[t][2].add(trace(T8, sc, o, d, 1));
                per[t][3].add(trace(T4, sc, o, d, 1));
            }
        });
    for (auto &x : th) x.join();
    for (int m = 0; m < 4; m++) {
        Counts c;
        for (int t = 0; t < n_threads; t++) c.add(per[t][m]);
        const double n = (double)n_rays;
        double *o = out + m * 8;
        o[0] = c.node_visits / n, o[1] = c.leaf_visits / n, o[2] = c.prim_tests / n, o[3] = c.pushes / n;
        o[4] = c.sp_sum / std::max(1.0, c.node_visits), o[5] = c.max_sp, o[6] = c.box_tests / n, o[7] = c.hits / n;
        for (int k = 0; k < 16; k++) out[32 + m * 16 + k] = c.sp_over[k] / std::max(1.0, c.node_visits);
    }
    return 0;
}
