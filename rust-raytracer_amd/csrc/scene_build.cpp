// scene_build.cpp -- host side of rt_scene_create.
//
// 1. expand the scene objects into primitives (ids in declaration order) with their
//    bounding boxes, transforming mesh vertices as Mesh::new does
//    (src/tracer/mesh.rs:31-73);
// 2. build the reference-shaped binary tree: Bvh::new (src/tracer/bvh.rs:39-117) over
//    each mesh's triangles, then over the world, a mesh acting as one leaf whose
//    box is its own root box (src/tracer/mesh.rs:90-97);
// 3. number the primitives in the order the reference's traversal reaches them
//    (right child first, src/tracer/bvh.rs:148-155);
// 4. collapse the binary tree into 4-wide nodes.  A child slot is either a binary
//    node (with that node's box) or a run of primitives that the reference tests
//    right after one box test (with exactly that box), so a ray reaches a primitive
//    here iff it reaches it in the reference: boxes of skipped interior nodes
//    contain their children's boxes and the slab test is monotone in the box.
//
// All arithmetic is float32 with no fused operations except where the reference's
// vector library fuses (dot / cross / mag_sq); compile with -ffp-contract=off.
#include "scene_build.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace rtcore {
namespace {

const float kPi = 3.14159265358979323846f;
const float kEps = 1.1920929e-7f;  // f32::EPSILON

inline float fmin_(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
inline float fmax_(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }

struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float dot(V3 a, V3 b) { return std::fmaf(a.x, b.x, std::fmaf(a.y, b.y, a.z * b.z)); }
inline V3 cross(V3 a, V3 b) {
    return v3(std::fmaf(a.y, b.z, -a.z * b.y), std::fmaf(a.z, b.x, -a.x * b.z), std::fmaf(a.x, b.y, -a.y * b.x));
}
inline V3 normalized(V3 a) {
    float r = 1.0f / std::sqrt(dot(a, a));
    return v3(a.x * r, a.y * r, a.z * r);
}

// ultraviolet::Rotor3 as the reference uses it (src/tracer/cube.rs:99, src/tracer/mesh.rs:31)
struct Rotor {
    float s, xy, xz, yz;
};
inline Rotor rotor_plane(float angle, float pxy, float pxz, float pyz) {
    float half = angle * 0.5f;
    float sn = std::sin(half), cs = std::cos(half);
    return Rotor{cs, pxy * -sn, pxz * -sn, pyz * -sn};
}
inline Rotor rotor_mul(const Rotor &a, const Rotor &q) {
    Rotor r;
    r.s = a.s * q.s - a.xy * q.xy - a.xz * q.xz - a.yz * q.yz;
    r.xy = a.xy * q.s + a.s * q.xy + a.yz * q.xz - a.xz * q.yz;
    r.xz = a.xz * q.s + a.s * q.xz - a.yz * q.xy + a.xy * q.yz;
    r.yz = a.yz * q.s + a.s * q.yz + a.xz * q.xy - a.xy * q.xz;
    return r;
}
// from_euler_angles(roll, pitch, yaw) = rot_xz(yaw) * rot_yz(pitch) * rot_xy(roll), then normalized()
inline Rotor rotor_from_euler(float roll, float pitch, float yaw) {
    Rotor r = rotor_mul(rotor_mul(rotor_plane(yaw, 0, 1, 0), rotor_plane(pitch, 0, 0, 1)), rotor_plane(roll, 1, 0, 0));
    float m2 = r.s * r.s + std::fmaf(r.xy, r.xy, std::fmaf(r.xz, r.xz, r.yz * r.yz));
    float inv = 1.0f / std::sqrt(m2);
    return Rotor{r.s * inv, r.xy * inv, r.xz * inv, r.yz * inv};
}
inline V3 rotate(const Rotor &r, V3 v) {
    float fx = r.s * v.x + r.xy * v.y + r.xz * v.z;
    float fy = r.s * v.y - r.xy * v.x + r.yz * v.z;
    float fz = r.s * v.z - r.xz * v.x - r.yz * v.y;
    float fw = r.xy * v.z - r.xz * v.y + r.yz * v.x;
    return v3(r.s * fx + r.xy * fy + r.xz * fz + r.yz * fw, r.s * fy - r.xy * fx - r.xz * fw + r.yz * fz,
              r.s * fz + r.xy * fw - r.xz * fx - r.yz * fy);
}

struct Box {
    float lo[3], hi[3];
};
// Aabb::surrounding_box (src/tracer/cube.rs:179-193)
inline Box surrounding(const Box &a, const Box &b) {
    Box r;
    for (int k = 0; k < 3; k++) {
        r.lo[k] = fmin_(a.lo[k], b.lo[k]) - kEps;
        r.hi[k] = fmax_(a.hi[k], b.hi[k]) + kEps;
    }
    return r;
}
inline float half_area(const Box &b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

inline uint32_t fbits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

const int32_t kNone = INT32_MIN;
// child encoding of the binary tree: >= 0 node index, < 0 (and != kNone) ~primitive
struct BNode {
    Box box;
    int32_t left, right;
};
struct Item {
    int32_t child;
    Box box;
};

struct Builder {
    std::vector<BNode> nodes;
    std::atomic<int64_t> next_node{0};
    std::atomic<int> spare_threads{0};
    std::atomic<bool> too_deep{false};

    bool take_thread() {
        int v = spare_threads.load();
        while (v > 0)
            if (spare_threads.compare_exchange_weak(v, v - 1)) return true;
        return false;
    }
    void give_thread() { spare_threads.fetch_add(1); }

    // Stable top-down merge sort (split at n/2, ties keep the left run first).  The
    // comparator above is not a strict weak order, so the permutation depends on the
    // algorithm; this one defines the tree.  It runs on 16-byte keys (what the comparator reads of a
    // box on one axis, and the item's position) that ping-pong between two buffers; the 28-byte
    // items are moved once at the end.
    // Bvh::box_compare (src/tracer/bvh.rs:120-139): a box that strictly contains the other on the axis sorts first,
    // otherwise the "centres" decide, and the centre is max + min / 2 as written there.
    struct Key {
        float lo, hi, c;  // c = hi + lo / 2
        uint32_t at;
    };
    static bool key_less(const Key &a, const Key &b) {  // box_compare(a, b, axis) == Less
        if (a.lo < b.lo && a.hi > b.hi) return true;
        if (b.lo < a.lo && b.hi > a.hi) return false;
        return a.c < b.c;
    }
    // sorts a[0, n) with b as the other buffer; the result is left in a (to_a) or in b
    void sort_keys(Key *a, Key *b, size_t n, bool to_a) {
        if (n < 2) {
            if (n == 1 && !to_a) b[0] = a[0];
            return;
        }
        size_t mid = n / 2;
        if (n > (1u << 15) && take_thread()) {
            std::thread t([&]() { sort_keys(a, b, mid, !to_a); });
            sort_keys(a + mid, b + mid, n - mid, !to_a);
            t.join();
            give_thread();
        } else {
            sort_keys(a, b, mid, !to_a);
            sort_keys(a + mid, b + mid, n - mid, !to_a);
        }
        const Key *src = to_a ? b : a;
        Key *dst = to_a ? a : b;
        size_t i = 0, j = mid, k = 0;
        while (i < mid && j < n) {
            if (key_less(src[j], src[i]))
                dst[k++] = src[j++];
            else
                dst[k++] = src[i++];
        }
        while (i < mid) dst[k++] = src[i++];
        while (j < n) dst[k++] = src[j++];
    }
    // buffers that go with an item array, indexed like it (a run keeps its offsets)
    struct Scratch {
        Item *tmp;
        Key *ka, *kb;
        Scratch at(size_t off) const { return Scratch{tmp + off, ka + off, kb + off}; }
    };
    void merge_sort(Item *a, Scratch sc, size_t n, int axis) {
        if (n < 2) return;
        Item *tmp = sc.tmp;
        Key *ka = sc.ka, *kb = sc.kb;
        for (size_t i = 0; i < n; i++) {
            const Box &b = a[i].box;
            ka[i] = Key{b.lo[axis], b.hi[axis], b.hi[axis] + b.lo[axis] * 0.5f, (uint32_t)i};
        }
        sort_keys(ka, kb, n, true);
        for (size_t i = 0; i < n; i++) tmp[i] = a[ka[i].at];
        std::memcpy(a, tmp, n * sizeof(Item));
    }

    // Bvh::largest_axis (src/tracer/bvh.rs:83-104): the fold pads by EPSILON at every step.
    static int largest_axis(const Item *it, size_t n) {
        Box b = it[0].box;
        for (size_t i = 0; i < n; i++) b = surrounding(b, it[i].box);
        float w = b.hi[0] - b.lo[0], h = b.hi[1] - b.lo[1], d = b.hi[2] - b.lo[2];
        if (w > h && w > d) return 0;
        if (h > d) return 1;
        return 2;
    }
    // Bvh::true_middle (src/tracer/bvh.rs:106-117)
    static size_t true_middle(const Item *it, size_t n, int axis) {
        size_t mid = 1;
        float mid_world = (it[0].box.lo[axis] + it[n - 1].box.hi[axis]) * 0.5f;
        while (mid < n - 1 && (it[mid].box.hi[axis] + it[mid].box.lo[axis]) * 0.5f <= mid_world) mid++;
        return mid;
    }
    const Box &child_box(int32_t c, const RawVector<Box> &prim_boxes) const {
        return c >= 0 ? nodes[c].box : prim_boxes[~c];
    }

    // Bvh::new (src/tracer/bvh.rs:39-81)
    int32_t build(Item *it, Scratch tmp, size_t n, const RawVector<Box> &prim_boxes, int depth) {
        if (depth > 4096) {
            too_deep = true;
            return kNone;
        }
        int axis = largest_axis(it, n);
        merge_sort(it, tmp, n, axis);
        int32_t me = (int32_t)next_node.fetch_add(1);
        int32_t l = kNone, r = kNone;
        if (n == 1) {
            l = it[0].child;
        } else if (n == 2) {
            l = it[0].child;
            r = it[1].child;
        } else if (n == 3) {
            l = build(it, tmp, 2, prim_boxes, depth + 1);
            r = it[2].child;
        } else {
            size_t mid = true_middle(it, n, axis);
            if (n > (1u << 14) && take_thread()) {
                std::thread t([&]() { l = build(it, tmp, mid, prim_boxes, depth + 1); });
                r = build(it + mid, tmp.at(mid), n - mid, prim_boxes, depth + 1);
                t.join();
                give_thread();
            } else {
                l = build(it, tmp, mid, prim_boxes, depth + 1);
                r = build(it + mid, tmp.at(mid), n - mid, prim_boxes, depth + 1);
            }
        }
        if (too_deep) return kNone;
        BNode &nd = nodes[me];
        nd.left = l;
        nd.right = r;
        if (l != kNone && r != kNone)
            nd.box = surrounding(child_box(l, prim_boxes), child_box(r, prim_boxes));
        else
            nd.box = child_box(l, prim_boxes);
        return me;
    }
};

struct Slot {
    Box box;
    int32_t child;        // binary-tree child encoding
    bool in_sub = false;  // the child lies in (or is the root of) one of the subtrees that are emitted on their own
};

// Where the wide nodes of one piece of the tree go.  The piece below the world's root is emitted first and leaves the
// subtrees of the meshes to tasks that run side by side, each into an emitter of its own; the pieces are then put
// one after the other and the node indices inside each piece moved by the piece's position.
struct Emitter {
    std::vector<float> out;  // 32 floats per wide node
    uint32_t depth_max = 0;
    bool overflow = false;
    struct Deferred {
        int32_t node;     // binary node that roots the wide node to emit
        uint32_t depth;
        size_t at;        // float index in `out` of the child word that names it
    };
    std::vector<Deferred> deferred;
};

struct Collapser {
    const std::vector<BNode> &bn;
    const std::vector<uint32_t> &rank_of;  // primitive -> visit rank
    std::vector<uint8_t> is_sub;           // per binary node: roots a subtree that is planned and emitted on its own

    Collapser(const std::vector<BNode> &b, const std::vector<uint32_t> &r) : bn(b), rank_of(r) {}

    bool terminal(int32_t b) const {
        const BNode &n = bn[b];
        return (n.left == kNone || n.left < 0) && (n.right == kNone || n.right < 0);
    }
    // the slots the reference reaches right after passing node b's box
    int child_slots(int32_t b, Slot *s) const {
        const BNode &n = bn[b];
        int c = 0;
        const int32_t ch[2] = {n.left, n.right};
        for (int k = 0; k < 2; k++) {
            if (ch[k] == kNone) continue;
            if (ch[k] >= 0)
                s[c++] = Slot{bn[ch[k]].box, ch[k]};
            else
                s[c++] = Slot{n.box, ch[k]};  // a bare leaf is tested under its parent's box
        }
        return c;
    }
    static uint32_t leaf_word(Emitter &e, uint32_t first, uint32_t count) {
        if (first > RT_LEAF_FIRST_MASK || count == 0 || count > 2) e.overflow = true;  // the kernels fetch two records per leaf
        return RT_CHILD_LEAF | ((count - 1) << 28) | first;
    }

    // Which binary nodes become roots of wide nodes.  A ray opens a wide node about as often as it
    // hits the root's box, so the cost of a collapse is taken as the sum of the half areas of the
    // wide nodes' root boxes (leaf slots are the same in every collapse), and the cheapest
    // collapse follows from a bottom-up table over the binary tree:
    //   dist(b, j)  the cheapest way to hand the two subtrees of b to at most j slots of one wide node
    //   root(b)     half_area(b) + dist(b, 4): b roots a wide node of its own
    // where a subtree given k slots is either one slot that roots its own wide node or is dissolved
    // further into at most k slots.
    std::vector<float> root_cost;
    std::vector<float> dist[3];      // [j - 2][node]
    std::vector<uint8_t> to_left[3]; // slots that go to the left child in dist(b, j)
    bool optimal = false;

    float subtree_cost(int32_t c, int k) const {  // subtree c in at most k slots
        if (c < 0 || terminal(c)) return 0.0f;
        float best = root_cost[c];
        for (int j = 2; j <= k; j++) best = std::min(best, dist[j - 2][c]);
        return best;
    }
    void plan_begin() {
        const size_t n = bn.size();
        root_cost.assign(n, 0.0f);
        for (int j = 0; j < 3; j++) dist[j].assign(n, 0.0f), to_left[j].assign(n, 1);
        optimal = true;
    }
    // Fills the table for the subtree of `root`; subtrees marked in is_sub (other than root itself) are taken as done.
    // Calls for disjoint subtrees may run side by side.
    void plan(int32_t root) {
        std::vector<int32_t> order;
        std::vector<int32_t> stack{root};
        while (!stack.empty()) {
            int32_t b = stack.back();
            stack.pop_back();
            if (terminal(b)) continue;
            if (b != root && !is_sub.empty() && is_sub[b]) continue;
            order.push_back(b);
            if (bn[b].left != kNone && bn[b].left >= 0) stack.push_back(bn[b].left);
            if (bn[b].right != kNone && bn[b].right >= 0) stack.push_back(bn[b].right);
        }
        for (size_t i = order.size(); i-- > 0;) {  // children before parents
            const int32_t b = order[i], l = bn[b].left, r = bn[b].right;
            for (int j = 2; j <= 4; j++) {
                float best;
                int best_k;
                if (l == kNone || r == kNone) {
                    best_k = (l == kNone) ? 0 : j;
                    best = subtree_cost(l == kNone ? r : l, j);
                } else {
                    best = INFINITY, best_k = 1;
                    for (int k = 1; k < j; k++) {
                        float v = subtree_cost(l, k) + subtree_cost(r, j - k);
                        if (v < best) best = v, best_k = k;
                    }
                }
                dist[j - 2][b] = best;
                to_left[j - 2][b] = (uint8_t)best_k;
            }
            root_cost[b] = half_area(bn[b].box) + dist[2][b];
        }
    }
    // the slots of the wide node that dissolves b into at most j slots, by the table
    void gather(int32_t b, int j, Slot *slots, int &n, bool in_sub) const {
        const int k = to_left[j - 2][b];
        place(b, bn[b].left, k, slots, n, in_sub);
        place(b, bn[b].right, j - k, slots, n, in_sub);
    }
    void place(int32_t parent, int32_t c, int k, Slot *slots, int &n, bool in_sub) const {
        if (c == kNone || k <= 0) return;
        if (c < 0) {
            slots[n++] = Slot{bn[parent].box, c, in_sub};  // a bare leaf is tested under its parent's box
            return;
        }
        if (!is_sub.empty() && is_sub[c]) in_sub = true;
        if (terminal(c)) {
            slots[n++] = Slot{bn[c].box, c, in_sub};
            return;
        }
        float best = root_cost[c];
        int best_j = 1;
        for (int j = 2; j <= k; j++)
            if (dist[j - 2][c] < best) best = dist[j - 2][c], best_j = j;
        if (best_j == 1)
            slots[n++] = Slot{bn[c].box, c, in_sub};
        else
            gather(c, best_j, slots, n, in_sub);
    }

    // Emits the wide node rooted at binary node b and everything below it into `e`; returns its index in e.out.
    // With `defer`, wide nodes that lie in one of the is_sub subtrees are not emitted but noted in e.deferred.
    uint32_t make(Emitter &e, int32_t b, uint32_t depth, bool defer) const {
        if (depth > e.depth_max) e.depth_max = depth;
        Slot slots[4];
        int n;
        if (terminal(b)) {
            slots[0] = Slot{bn[b].box, b};
            n = 1;
        } else if (optimal) {
            n = 0;
            gather(b, 4, slots, n, false);
        } else {
            n = child_slots(b, slots);
            for (;;) {
                int best = -1;
                float best_area = -1.0f;
                for (int i = 0; i < n; i++) {
                    int32_t c = slots[i].child;
                    if (c < 0 || terminal(c)) continue;
                    int kids = (bn[c].left != kNone) + (bn[c].right != kNone);
                    if (n - 1 + kids > 4) continue;
                    float a = half_area(slots[i].box);
                    if (a > best_area) {
                        best_area = a;
                        best = i;
                    }
                }
                if (best < 0) break;
                Slot kids[2];
                int k = child_slots(slots[best].child, kids);
                slots[best] = kids[0];
                if (k > 1) slots[n++] = kids[1];
            }
        }
        const size_t me = e.out.size() / 32;
        e.out.resize(e.out.size() + 32, 0.0f);
        uint32_t words[4] = {RT_CHILD_NONE, RT_CHILD_NONE, RT_CHILD_NONE, RT_CHILD_NONE};
        float boxes[6][4];
        for (int a = 0; a < 6; a++)
            for (int i = 0; i < 4; i++) boxes[a][i] = INFINITY;  // an empty slot: a box no ray hits (see node_children_sorted)
        for (int i = 0; i < n; i++) {
            for (int a = 0; a < 3; a++) {
                boxes[a][i] = slots[i].box.lo[a];
                boxes[3 + a][i] = slots[i].box.hi[a];
            }
            int32_t c = slots[i].child;
            if (c < 0) {
                words[i] = leaf_word(e, rank_of[~c], 1);
            } else if (terminal(c)) {
                const BNode &t = bn[c];
                // right is visited first, so it has the lower rank
                if (t.left != kNone && t.right != kNone) {
                    uint32_t r0 = rank_of[~t.right], r1 = rank_of[~t.left];
                    if (r1 != r0 + 1) e.overflow = true;
                    words[i] = leaf_word(e, r0, 2);
                } else {
                    words[i] = leaf_word(e, rank_of[~(t.left != kNone ? t.left : t.right)], 1);
                }
            } else if (defer && optimal && (slots[i].in_sub || (!is_sub.empty() && is_sub[c]))) {
                e.deferred.push_back(Emitter::Deferred{c, depth + 1, me * 32 + 24 + (size_t)i});
                words[i] = 0;  // set when the pieces are put together
            } else {
                words[i] = make(e, c, depth + 1, defer);
            }
        }
        float *dst = &e.out[me * 32];
        for (int a = 0; a < 6; a++)
            for (int i = 0; i < 4; i++) dst[a * 4 + i] = boxes[a][i];
        std::memcpy(dst + 24, words, 16);
        return (uint32_t)me;
    }
};


// ---------------------------------------------------------------------------
// Interior rebuild.  What the reference fixes is (a) the order in which primitives
// are visited and (b) the box that is tested last before each primitive ("guard
// box").  Every box above a guard box only has to contain it, so the interior of
// the tree can be rebuilt for traversal speed: the guard boxes with their 1-2
// primitives become the leaves of a binned-SAH binary tree, which is then
// collapsed into 4-wide nodes whose interior boxes are unions of guard boxes.
// ---------------------------------------------------------------------------
struct LeafSlot {
    Box box;
    uint32_t first, count;
};

void collect_leaf_slots(const std::vector<BNode> &bn, int32_t root, const std::vector<uint32_t> &rank_of,
                        std::vector<LeafSlot> &out) {
    std::vector<int32_t> stack;
    stack.push_back(root);
    while (!stack.empty()) {
        int32_t b = stack.back();
        stack.pop_back();
        const BNode &n = bn[b];
        bool l_prim = n.left != kNone && n.left < 0, r_prim = n.right != kNone && n.right < 0;
        bool l_node = n.left != kNone && n.left >= 0, r_node = n.right != kNone && n.right >= 0;
        if (!l_node && !r_node) {  // terminal: its primitives share this node's box
            if (l_prim && r_prim) {
                uint32_t r0 = rank_of[~n.right];  // right is visited first
                out.push_back(LeafSlot{n.box, r0, 2});
            } else {
                out.push_back(LeafSlot{n.box, rank_of[~(l_prim ? n.left : n.right)], 1});
            }
            continue;
        }
        if (l_prim) out.push_back(LeafSlot{n.box, rank_of[~n.left], 1});
        if (r_prim) out.push_back(LeafSlot{n.box, rank_of[~n.right], 1});
        if (l_node) stack.push_back(n.left);
        if (r_node) stack.push_back(n.right);
    }
}

struct SahNode {
    Box box;
    int32_t left, right;  // interior: node indices; leaf: left = ~slot, right = kNone
};

inline Box box_union(const Box &a, const Box &b) {
    Box r;
    for (int k = 0; k < 3; k++) r.lo[k] = fmin_(a.lo[k], b.lo[k]), r.hi[k] = fmax_(a.hi[k], b.hi[k]);
    return r;
}
inline Box empty_box() {
    Box b;
    for (int k = 0; k < 3; k++) b.lo[k] = INFINITY, b.hi[k] = -INFINITY;
    return b;
}

struct SahBuilder {
    const std::vector<LeafSlot> &slots;
    std::vector<SahNode> nodes;
    std::atomic<int64_t> next{0};
    std::atomic<int> spare_threads{0};
    static constexpr int kBins = 16;

    explicit SahBuilder(const std::vector<LeafSlot> &s) : slots(s) { nodes.resize(2 * s.size() + 2); }
    bool take_thread() {
        int v = spare_threads.load();
        while (v > 0)
            if (spare_threads.compare_exchange_weak(v, v - 1)) return true;
        return false;
    }
    int32_t build(uint32_t *idx, size_t n) {
        int32_t me = (int32_t)next.fetch_add(1);
        if (n == 1) {
            nodes[me] = SahNode{slots[idx[0]].box, ~(int32_t)idx[0], kNone};
            return me;
        }
        Box bounds = empty_box(), cb = empty_box();
        for (size_t i = 0; i < n; i++) {
            const Box &b = slots[idx[i]].box;
            bounds = box_union(bounds, b);
            for (int k = 0; k < 3; k++) {
                float c = 0.5f * (b.lo[k] + b.hi[k]);
                cb.lo[k] = fmin_(cb.lo[k], c), cb.hi[k] = fmax_(cb.hi[k], c);
            }
        }
        int best_axis = -1, best_bin = 0;
        float best_cost = INFINITY;
        for (int axis = 0; axis < 3; axis++) {
            float ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 0.0f)) continue;
            float scale = (float)kBins / ext;
            Box bin_box[kBins];
            size_t bin_n[kBins];
            for (int b = 0; b < kBins; b++) bin_box[b] = empty_box(), bin_n[b] = 0;
            for (size_t i = 0; i < n; i++) {
                const Box &b = slots[idx[i]].box;
                int k = (int)((0.5f * (b.lo[axis] + b.hi[axis]) - cb.lo[axis]) * scale);
                if (k >= kBins) k = kBins - 1;
                if (k < 0) k = 0;
                bin_box[k] = box_union(bin_box[k], b);
                bin_n[k]++;
            }
            float right_area[kBins];
            size_t right_n[kBins];
            Box acc = empty_box();
            size_t cnt = 0;
            for (int b = kBins - 1; b > 0; b--) {
                acc = box_union(acc, bin_box[b]);
                cnt += bin_n[b];
                right_area[b] = cnt ? half_area(acc) : 0.0f;
                right_n[b] = cnt;
            }
            acc = empty_box();
            cnt = 0;
            for (int b = 0; b < kBins - 1; b++) {
                acc = box_union(acc, bin_box[b]);
                cnt += bin_n[b];
                if (cnt == 0 || right_n[b + 1] == 0) continue;
                float cost = half_area(acc) * (float)cnt + right_area[b + 1] * (float)right_n[b + 1];
                if (cost < best_cost) best_cost = cost, best_axis = axis, best_bin = b;
            }
        }
        size_t mid;
        if (best_axis < 0) {
            mid = n / 2;  // all centroids coincide
        } else {
            float ext = cb.hi[best_axis] - cb.lo[best_axis];
            float scale = (float)kBins / ext;
            auto bin_of = [&](uint32_t s) {
                const Box &b = slots[s].box;
                int k = (int)((0.5f * (b.lo[best_axis] + b.hi[best_axis]) - cb.lo[best_axis]) * scale);
                if (k >= kBins) k = kBins - 1;
                if (k < 0) k = 0;
                return k;
            };
            mid = (size_t)(std::stable_partition(idx, idx + n, [&](uint32_t s) { return bin_of(s) <= best_bin; }) - idx);
            if (mid == 0 || mid == n) mid = n / 2;
        }
        int32_t l, r;
        if (n > (1u << 15) && take_thread()) {
            std::thread t([&]() { l = build(idx, mid); });
            r = build(idx + mid, n - mid);
            t.join();
            spare_threads.fetch_add(1);
        } else {
            l = build(idx, mid);
            r = build(idx + mid, n - mid);
        }
        nodes[me] = SahNode{bounds, l, r};
        return me;
    }
};

struct SahCollapser {
    const std::vector<SahNode> &sn;
    const std::vector<LeafSlot> &slots;
    RawVector<float> &out;
    uint32_t depth_max = 0;
    bool overflow = false;
    SahCollapser(const std::vector<SahNode> &n, const std::vector<LeafSlot> &s, RawVector<float> &o) : sn(n), slots(s), out(o) {}
    bool is_leaf(int32_t n) const { return sn[n].left < 0; }
    uint32_t leaf_word(const LeafSlot &s) {
        if (s.first > RT_LEAF_FIRST_MASK || s.count == 0 || s.count > 2) overflow = true;  // the kernels fetch two records per leaf
        return RT_CHILD_LEAF | ((s.count - 1) << 28) | s.first;
    }
    uint32_t make(int32_t root, uint32_t depth) {
        if (depth > depth_max) depth_max = depth;
        int32_t kids[4];
        int n = 0;
        if (is_leaf(root)) {
            kids[n++] = root;
        } else {
            kids[n++] = sn[root].left;
            kids[n++] = sn[root].right;
            while (n < 4) {
                int best = -1;
                float best_area = -1.0f;
                for (int i = 0; i < n; i++) {
                    if (is_leaf(kids[i])) continue;
                    float a = half_area(sn[kids[i]].box);
                    if (a > best_area) best_area = a, best = i;
                }
                if (best < 0) break;
                int32_t k = kids[best];
                kids[best] = sn[k].left;
                kids[n++] = sn[k].right;
            }
        }
        size_t me = out.size() / 32;
        out.resize(out.size() + 32, 0.0f);
        uint32_t words[4] = {RT_CHILD_NONE, RT_CHILD_NONE, RT_CHILD_NONE, RT_CHILD_NONE};
        float boxes[6][4];
        for (int a = 0; a < 6; a++)
            for (int i = 0; i < 4; i++) boxes[a][i] = INFINITY;  // an empty slot: a box no ray hits
        for (int i = 0; i < n; i++) {
            const Box &b = sn[kids[i]].box;
            for (int a = 0; a < 3; a++) boxes[a][i] = b.lo[a], boxes[3 + a][i] = b.hi[a];
            words[i] = is_leaf(kids[i]) ? leaf_word(slots[~sn[kids[i]].left]) : make(kids[i], depth + 1);
        }
        float *dst = &out[me * 32];
        for (int a = 0; a < 6; a++)
            for (int i = 0; i < 4; i++) dst[a * 4 + i] = boxes[a][i];
        std::memcpy(dst + 24, words, 16);
        return (uint32_t)me;
    }
};

}  // namespace

void make_camera(const RtCamera &c, float aspect, RtDevCamera &o) {
    V3 eye = v3(c.eye[0], c.eye[1], c.eye[2]), lookat = v3(c.lookat[0], c.lookat[1], c.lookat[2]);
    V3 vup = v3(c.vup[0], c.vup[1], c.vup[2]);
    float h = std::tan((c.fov_degrees * (kPi / 180.0f)) / 2.0f);
    float vh = 2.0f * h;
    float vw = aspect * vh;
    V3 w = normalized(sub(eye, lookat));
    V3 u = normalized(cross(vup, w));
    V3 v = cross(w, u);
    float fw = c.focus_dist * vw, fh = c.focus_dist * vh;
    V3 hor = v3(fw * u.x, fw * u.y, fw * u.z);
    V3 ver = v3(fh * v.x, fh * v.y, fh * v.z);
    V3 fwv = v3(c.focus_dist * w.x, c.focus_dist * w.y, c.focus_dist * w.z);
    V3 llc = sub(sub(sub(eye, v3(hor.x / 2.0f, hor.y / 2.0f, hor.z / 2.0f)), v3(ver.x / 2.0f, ver.y / 2.0f, ver.z / 2.0f)), fwv);
    std::memset(&o, 0, sizeof(o));
    o.eye[0] = eye.x, o.eye[1] = eye.y, o.eye[2] = eye.z;
    o.lens_radius = c.aperture / 2.0f;
    o.horizontal[0] = hor.x, o.horizontal[1] = hor.y, o.horizontal[2] = hor.z;
    o.vertical[0] = ver.x, o.vertical[1] = ver.y, o.vertical[2] = ver.z;
    o.llc[0] = llc.x, o.llc[1] = llc.y, o.llc[2] = llc.z;
    o.u[0] = u.x, o.u[1] = u.y, o.u[2] = u.z;
    o.v[0] = v.x, o.v[1] = v.y, o.v[2] = v.z;
}

// fn(lo, hi) over [0, n) on up to 16 host threads (one when n is small)
template <class F>
void parallel_ranges(uint64_t n, F fn) {
    unsigned hw = std::thread::hardware_concurrency();
    uint64_t parts = std::min<uint64_t>(std::min<unsigned>(hw ? hw : 1u, 16u), n / 65536u);
    if (parts <= 1) {
        fn((uint64_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    for (uint64_t k = 0; k < parts; k++)
        pool.emplace_back([=, &fn]() { fn(n * k / parts, n * (k + 1) / parts); });
    for (auto &t : pool) t.join();
}

// Validates a scene description and counts its primitives (one per object, one per face of a mesh object).
int check_scene_desc(const RtSceneDesc *d, uint64_t &total_out, std::string &err, bool check_indices) {
    if (!d || (d->n_objects && !d->objects) || (d->n_materials && !d->materials) || (d->n_meshes && !d->meshes)) {
        err = "null scene description";
        return RT_ERR_INVALID_ARGUMENT;
    }
    if (d->n_objects == 0) {
        err = "a scene needs at least one object (the reference panics on an empty world)";
        return RT_ERR_EMPTY_SCENE;
    }
    // ---- count primitives
    uint64_t total = 0;
    for (uint64_t i = 0; i < d->n_objects; i++) {
        const RtObject &o = d->objects[i];
        if (o.material < 0 || (uint64_t)o.material >= d->n_materials) {
            err = "object " + std::to_string(i) + ": material index out of range";
            return RT_ERR_INVALID_ARGUMENT;
        }
        if (o.type == RT_OBJ_MESH) {
            if (o.mesh < 0 || (uint64_t)o.mesh >= d->n_meshes) {
                err = "object " + std::to_string(i) + ": mesh index out of range";
                return RT_ERR_INVALID_ARGUMENT;
            }
            const RtMesh &m = d->meshes[o.mesh];
            if (m.n_faces == 0) {
                err = "object " + std::to_string(i) + ": mesh without faces";
                return RT_ERR_EMPTY_SCENE;
            }
            if (!m.positions || !m.normals || !m.indices) {
                err = "object " + std::to_string(i) + ": mesh needs positions, normals and indices";
                return RT_ERR_INVALID_ARGUMENT;
            }
            for (uint64_t k = 0; check_indices && k < 3 * m.n_faces; k++)
                if (m.indices[k] >= m.n_vertices) {
                    err = "object " + std::to_string(i) + ": vertex index out of range";
                    return RT_ERR_INVALID_ARGUMENT;
                }
            total += m.n_faces;
        } else if (o.type >= RT_OBJ_SPHERE && o.type <= RT_OBJ_TRIANGLE) {
            total += 1;
        } else {
            err = "object " + std::to_string(i) + ": unknown type";
            return RT_ERR_INVALID_ARGUMENT;
        }
    }
    if (total > RT_LEAF_FIRST_MASK) {
        err = "too many primitives";
        return RT_ERR_INVALID_ARGUMENT;
    }
    if (d->n_materials >= (1u << 28)) {
        err = "too many materials";
        return RT_ERR_INVALID_ARGUMENT;
    }

    total_out = total;
    return RT_OK;
}

// The record, the normals word block and the bounding box of a scene object that is ONE primitive (sphere, axis box,
// rotated box, triangle): rec[12] words, nrm[12] floats (three vertex normals, 4 floats apart; zero for non-triangles),
// box = lo[3], hi[3].  `id` is the primitive id (declaration order).  Mesh faces are made by triangle_record.
void triangle_record(const float vv[9], const float nn[9], bool two_sided, uint32_t mat_bits, uint32_t id, uint32_t *rec,
                     float *nrm, float *box) {
    for (int k = 0; k < 3; k++) {
        rec[k] = fbits(vv[k]);
        rec[3 + k] = fbits(vv[3 + k] - vv[k]);
        rec[6 + k] = fbits(vv[6 + k] - vv[k]);
    }
    rec[9] = 0;
    rec[10] = RT_PRIM_TRIANGLE | (two_sided ? RT_PRIM_FLAG : 0) | mat_bits;
    rec[11] = id;
    for (int k = 0; k < 3; k++) nrm[4 * k] = nn[3 * k], nrm[4 * k + 1] = nn[3 * k + 1], nrm[4 * k + 2] = nn[3 * k + 2], nrm[4 * k + 3] = 0.0f;
    for (int a = 0; a < 3; a++) {  // Triangle::bounding_box (src/tracer/triangle.rs:80-105)
        box[a] = fmin_(vv[a], fmin_(vv[3 + a], vv[6 + a]));
        box[3 + a] = fmax_(vv[a], fmax_(vv[3 + a], vv[6 + a]));
    }
}

void simple_object_record(const RtObject &o, uint32_t id, uint32_t *rec, float *nrm, float *box) {
    const uint32_t mat = (uint32_t)o.material << RT_PRIM_MAT_SHIFT;
    for (int k = 0; k < 12; k++) rec[k] = 0, nrm[k] = 0.0f;
    rec[11] = id;
    switch (o.type) {
        case RT_OBJ_SPHERE: {
            for (int k = 0; k < 3; k++) rec[k] = fbits(o.a[k]);
            rec[3] = fbits(o.b[0]);
            rec[10] = RT_PRIM_SPHERE | mat;
            float r = 1.0f * std::fabs(o.b[0]);  // Sphere::bounding_box (src/tracer/sphere.rs:60-65)
            for (int k = 0; k < 3; k++) box[k] = o.a[k] - r, box[3 + k] = o.a[k] + r;
            break;
        }
        case RT_OBJ_ABOX:
        case RT_OBJ_CUBE: {
            // ABox::new (src/tracer/cube.rs:20-38)
            bool hollow = fmin_(fmin_(o.b[0], o.b[1]), o.b[2]) < 0.0f;
            float half[3], lo[3], hi[3];
            for (int k = 0; k < 3; k++) {
                half[k] = std::fabs(o.b[k] * 0.5f);
                lo[k] = o.a[k] - half[k];
                hi[k] = o.a[k] + half[k];
            }
            if (o.type == RT_OBJ_ABOX) {
                for (int k = 0; k < 3; k++) rec[k] = fbits(lo[k]), rec[3 + k] = fbits(hi[k]);
                rec[10] = RT_PRIM_ABOX | (hollow ? RT_PRIM_FLAG : 0) | mat;
                for (int k = 0; k < 3; k++) box[k] = lo[k], box[3 + k] = hi[k];
            } else {
                // Cube::new / Cube::bounding_box (src/tracer/cube.rs:90-102, :120-148)
                Rotor rot = rotor_from_euler(o.c[2] * kPi, o.c[0] * kPi, o.c[1] * kPi);
                for (int k = 0; k < 3; k++) rec[k] = fbits(o.a[k]), rec[3 + k] = fbits(half[k]);
                rec[6] = fbits(rot.s), rec[7] = fbits(rot.xy), rec[8] = fbits(rot.xz), rec[9] = fbits(rot.yz);
                rec[10] = RT_PRIM_CUBE | (hollow ? RT_PRIM_FLAG : 0) | mat;
                V3 c = v3(o.a[0], o.a[1], o.a[2]);
                for (int k = 0; k < 3; k++) box[k] = INFINITY, box[3 + k] = -INFINITY;
                for (int corner = 0; corner < 8; corner++) {
                    V3 p = v3((corner & 1) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 4) ? hi[2] : lo[2]);
                    V3 q = add(rotate(rot, sub(p, c)), c);
                    box[0] = fmin_(box[0], q.x), box[1] = fmin_(box[1], q.y), box[2] = fmin_(box[2], q.z);
                    box[3] = fmax_(box[3], q.x), box[4] = fmax_(box[4], q.y), box[5] = fmax_(box[5], q.z);
                }
            }
            break;
        }
        default:  // RT_OBJ_TRIANGLE
            triangle_record(o.tri_vertices, o.tri_normals, o.flag != 0, mat, id, rec, nrm, box);
            break;
    }
}

// The rotor of Mesh::new (src/tracer/mesh.rs:31): rotation in radians, from_euler_angles(z, x, y), normalised.
void mesh_rotor(const RtMesh &m, float rot[4]) {
    Rotor r = rotor_from_euler(m.rotation[2], m.rotation[0], m.rotation[1]);
    rot[0] = r.s, rot[1] = r.xy, rot[2] = r.xz, rot[3] = r.yz;
}

int build_host_scene(const RtSceneDesc *d, HostScene &hs, std::string &err) {
    const bool debug_tree = std::getenv("RTCORE_DEBUG_TREE") != nullptr;
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char *name) {  // RTCORE_DEBUG_TREE=1: where the build time goes
        auto now = std::chrono::steady_clock::now();
        if (debug_tree) std::fprintf(stderr, "[rtcore] build: %-28s %.3f s\n", name, std::chrono::duration<double>(now - t_phase).count());
        t_phase = now;
    };
    uint64_t total = 0;
    {
        int rc = check_scene_desc(d, total, err);
        if (rc) return rc;
    }

    phase("checks");
    // ---- primitives in declaration order
    RawVector<uint32_t> rec(total * 12);  // (every word is written by the record functions)
    RawVector<float> nrm(total * 12);
    RawVector<Box> boxes(total);
    struct MeshSpan {
        uint64_t first, count;
    };
    struct MeshJob {
        uint64_t object, first;  // scene object, id of its first face
    };
    std::vector<MeshJob> mesh_jobs;
    std::vector<Item> world_items;
    std::vector<MeshSpan> mesh_spans;
    std::vector<size_t> mesh_world_slot;
    uint64_t id = 0;
    hs.n_triangles = 0;
    for (uint64_t i = 0; i < d->n_objects; i++) {
        const RtObject &o = d->objects[i];
        switch (o.type) {
            case RT_OBJ_SPHERE:
            case RT_OBJ_ABOX:
            case RT_OBJ_CUBE:
            case RT_OBJ_TRIANGLE: {
                float bx[6];
                simple_object_record(o, (uint32_t)id, &rec[id * 12], &nrm[id * 12], bx);
                for (int k = 0; k < 3; k++) boxes[id].lo[k] = bx[k], boxes[id].hi[k] = bx[3 + k];
                world_items.push_back(Item{~(int32_t)id, boxes[id]});
                if (o.type == RT_OBJ_TRIANGLE) hs.n_triangles++;
                id++;
                break;
            }
            default: {  // RT_OBJ_MESH: its faces are made below, side by side with the other meshes
                const RtMesh &m = d->meshes[o.mesh];
                mesh_jobs.push_back(MeshJob{i, id});
                mesh_spans.push_back(MeshSpan{id, m.n_faces});
                mesh_world_slot.push_back(world_items.size());
                world_items.push_back(Item{kNone, Box{}});  // filled once the mesh tree exists
                hs.n_triangles += m.n_faces;
                id += m.n_faces;
                break;
            }
        }
    }
    {
        // Mesh::new (src/tracer/mesh.rs:22-87) for faces [f_lo, f_hi) of one mesh object
        auto mesh_faces = [&](const MeshJob &job, uint64_t f_lo, uint64_t f_hi) {
            const RtObject &o = d->objects[job.object];
            const RtMesh &m = d->meshes[o.mesh];
            const uint32_t mat = (uint32_t)o.material << RT_PRIM_MAT_SHIFT;
            Rotor rot = rotor_from_euler(m.rotation[2], m.rotation[0], m.rotation[1]);
            V3 sc = v3(m.scale[0], m.scale[1], m.scale[2]);
            V3 tr = v3(m.translation[0], m.translation[1], m.translation[2]);
            for (uint64_t f = f_lo; f < f_hi; f++) {
                float vv[9], nn[9], bx[6];
                for (int k = 0; k < 3; k++) {
                    uint64_t vi = m.indices[3 * f + k];
                    V3 p = v3(m.positions[3 * vi], m.positions[3 * vi + 1], m.positions[3 * vi + 2]);
                    V3 q = add(rotate(rot, mul(p, sc)), tr);
                    V3 n = rotate(rot, v3(m.normals[3 * vi], m.normals[3 * vi + 1], m.normals[3 * vi + 2]));
                    vv[3 * k] = q.x, vv[3 * k + 1] = q.y, vv[3 * k + 2] = q.z;
                    nn[3 * k] = n.x, nn[3 * k + 1] = n.y, nn[3 * k + 2] = n.z;
                }
                const uint64_t p = job.first + f;
                triangle_record(vv, nn, !m.cull_backface, mat, (uint32_t)p, &rec[p * 12], &nrm[p * 12], bx);
                for (int k = 0; k < 3; k++) boxes[p].lo[k] = bx[k], boxes[p].hi[k] = bx[3 + k];
            }
        };
        unsigned hw0 = std::thread::hardware_concurrency();
        hw0 = hw0 ? std::min(hw0, 16u) : 1u;
        if (mesh_jobs.size() < hw0) {  // few meshes: the threads share each mesh's faces
            for (const MeshJob &job : mesh_jobs)
                parallel_ranges(d->meshes[d->objects[job.object].mesh].n_faces,
                                [&](uint64_t f_lo, uint64_t f_hi) { mesh_faces(job, f_lo, f_hi); });
        } else {  // many meshes: one mesh at a time per thread
            std::atomic<size_t> next_job{0};
            auto worker = [&]() {
                for (;;) {
                    size_t j = next_job.fetch_add(1);
                    if (j >= mesh_jobs.size()) break;
                    mesh_faces(mesh_jobs[j], 0, d->meshes[d->objects[mesh_jobs[j].object].mesh].n_faces);
                }
            };
            std::vector<std::thread> pool;
            for (unsigned w = 1; w < hw0; w++) pool.emplace_back(worker);
            worker();
            for (auto &t : pool) t.join();
        }
    }

    phase("primitive records and boxes");
    // ---- binary trees
    Builder B;
    B.nodes.resize(2 * (total + world_items.size()) + 2);  // a run of n items makes at most 2n - 1 nodes
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    if (hw > 16) hw = 16;
    B.spare_threads = (int)hw - 1;
    {
        uint64_t biggest = 0;
        for (auto &s : mesh_spans) biggest = std::max(biggest, s.count);
        // one tree per mesh (src/tracer/mesh.rs:75-83); the meshes are independent, so they are built side by side
        // (a mesh large enough to use the threads itself takes them from the same pool)
        std::atomic<size_t> next_mesh{0};
        std::atomic<bool> failed{false};
        auto mesh_worker = [&]() {
            std::vector<Item> items(biggest), tmp(biggest);
            std::vector<Builder::Key> ka(biggest), kb(biggest);
            for (;;) {
                size_t mi = next_mesh.fetch_add(1);
                if (mi >= mesh_spans.size() || failed) break;
                const MeshSpan &s = mesh_spans[mi];
                for (uint64_t f = 0; f < s.count; f++) items[f] = Item{~(int32_t)(s.first + f), boxes[s.first + f]};
                int32_t root = B.build(items.data(), Builder::Scratch{tmp.data(), ka.data(), kb.data()}, s.count, boxes, 0);
                if (root == kNone || B.too_deep) {
                    failed = true;
                    break;
                }
                world_items[mesh_world_slot[mi]] = Item{root, B.nodes[root].box};
            }
        };
        {
            // small meshes: one worker per thread; large ones: few workers, the rest of the pool serves the sorts
            size_t workers = biggest > (1u << 20) ? 1 : std::min<size_t>(hw, mesh_spans.size());
            std::vector<std::thread> pool;
            if (workers > 1) B.spare_threads = (int)hw - (int)workers;
            for (size_t w = 1; w < workers; w++) pool.emplace_back(mesh_worker);
            if (!mesh_spans.empty()) mesh_worker();
            for (auto &t : pool) t.join();
            B.spare_threads = (int)hw - 1;
        }
        if (failed) {
            err = "mesh geometry too degenerate for the reference's median split";
            return RT_ERR_INTERNAL;
        }
        std::vector<Item> wtmp(world_items.size());
        std::vector<Builder::Key> wka(world_items.size()), wkb(world_items.size());
        int32_t root = B.build(world_items.data(), Builder::Scratch{wtmp.data(), wka.data(), wkb.data()}, world_items.size(), boxes, 0);
        if (root == kNone || B.too_deep) {
            err = "scene too degenerate for the reference's median split";
            return RT_ERR_INTERNAL;
        }
        hs.n_binary_nodes = (uint64_t)B.next_node.load();
        B.nodes.resize(hs.n_binary_nodes);

        phase("binary trees (sort + split)");
        // ---- subtrees that the passes below work on side by side: the nodes ten levels below the root
        std::vector<int32_t> subs;
        std::vector<uint8_t> is_sub(B.nodes.size(), 0);
        {
            std::vector<std::pair<int32_t, int>> stack{{root, 0}};
            while (!stack.empty()) {
                auto [c, depth] = stack.back();
                stack.pop_back();
                if (c == kNone || c < 0) continue;
                if (depth == 10) {
                    subs.push_back(c);
                    is_sub[c] = 1;
                    continue;
                }
                stack.push_back({B.nodes[c].left, depth + 1});
                stack.push_back({B.nodes[c].right, depth + 1});
            }
        }
        auto for_each_sub = [&](auto fn) {  // fn(index into subs), on up to hw threads
            std::atomic<size_t> next_sub{0};
            auto worker = [&]() {
                for (;;) {
                    size_t i = next_sub.fetch_add(1);
                    if (i >= subs.size()) break;
                    fn(i);
                }
            };
            std::vector<std::thread> pool;
            for (unsigned w = 1; w < std::min<size_t>(hw, subs.size()); w++) pool.emplace_back(worker);
            worker();
            for (auto &t : pool) t.join();
        };
        // ---- visit ranks: right child first.  Primitives below a subtree take consecutive ranks, so the subtrees are
        // counted side by side, given their first rank by a walk over the top of the tree, and numbered side by side.
        std::vector<uint32_t> rank_of(total, 0);
        {
            std::vector<uint32_t> sub_count(subs.size(), 0), sub_first(subs.size(), 0);
            std::vector<int32_t> sub_index(B.nodes.size(), -1);
            for (size_t i = 0; i < subs.size(); i++) sub_index[subs[i]] = (int32_t)i;
            auto walk = [&](int32_t from, uint32_t next, bool assign, bool top) -> uint32_t {
                std::vector<int32_t> stack{from};
                while (!stack.empty()) {
                    int32_t c = stack.back();
                    stack.pop_back();
                    if (c == kNone) continue;
                    if (c < 0) {
                        if (assign) rank_of[~c] = next;
                        next++;
                        continue;
                    }
                    if (top && is_sub[c]) {
                        sub_first[sub_index[c]] = next;
                        next += sub_count[sub_index[c]];
                        continue;
                    }
                    stack.push_back(B.nodes[c].left);   // popped second
                    stack.push_back(B.nodes[c].right);  // popped first
                }
                return next;
            };
            for_each_sub([&](size_t i) { sub_count[i] = walk(subs[i], 0, false, false); });
            const uint32_t covered = walk(root, 0, true, true);
            for_each_sub([&](size_t i) { walk(subs[i], sub_first[i], true, false); });
            if (covered != total) {
                err = "internal: visit order does not cover every primitive";
                return RT_ERR_INTERNAL;
            }
        }

        phase("visit ranks");
        // ---- wide nodes
        hs.nodes.clear();
        // Default: keep the reference's own interior topology (a spatial-median tree, which on evenly
        // tessellated meshes traverses as well as the SAH rebuild: 21.4 vs 24.7 node visits per ray on
        // the 1 M-triangle bench scene).  RTCORE_INTERIOR=sah selects the rebuild; results are
        // identical either way, only the number of box tests differs.
        const char *interior = experiment_env("RTCORE_INTERIOR");
        if (!(interior && std::strcmp(interior, "sah") == 0)) {
            Collapser C(B.nodes, rank_of);
            // RTCORE_COLLAPSE=greedy keeps the simpler rule (dissolve the child with the largest box first)
            const char *collapse = experiment_env("RTCORE_COLLAPSE");
            const bool greedy = collapse && std::strcmp(collapse, "greedy") == 0;
            if (!greedy) {
                // the table of every subtree, side by side, then of the top of the tree
                C.is_sub = is_sub;
                C.plan_begin();
                for_each_sub([&](size_t i) { C.plan(subs[i]); });
                C.plan(root);
            }
            // the top of the tree, then the pieces it left to the subtrees, side by side, each into its own emitter
            Emitter top;
            C.make(top, root, 1, !greedy);
            std::vector<Emitter> pieces(top.deferred.size());
            {
                std::atomic<size_t> next_piece{0};
                auto worker = [&]() {
                    for (;;) {
                        size_t i = next_piece.fetch_add(1);
                        if (i >= pieces.size()) break;
                        C.make(pieces[i], top.deferred[i].node, top.deferred[i].depth, false);
                    }
                };
                std::vector<std::thread> pool;
                for (unsigned w = 1; w < std::min<size_t>(hw, pieces.size()); w++) pool.emplace_back(worker);
                worker();
                for (auto &t : pool) t.join();
            }
            // put together: the top first, the pieces after it; node indices inside a piece move by its position
            std::vector<size_t> base(pieces.size() + 1, 0);
            base[0] = top.out.size() / 32;
            bool overflow = top.overflow;
            uint32_t depth_max = top.depth_max;
            for (size_t i = 0; i < pieces.size(); i++) {
                base[i + 1] = base[i] + pieces[i].out.size() / 32;
                overflow = overflow || pieces[i].overflow;
                depth_max = std::max(depth_max, pieces[i].depth_max);
            }
            if (overflow) {
                err = "internal: leaf encoding overflow";
                return RT_ERR_INTERNAL;
            }
            if (base.back() >= (1ull << 25)) {  // kernels address nodes with 32-bit byte offsets
                err = "too many tree nodes";
                return RT_ERR_INVALID_ARGUMENT;
            }
            hs.nodes.resize(base.back() * 32);
            std::memcpy(hs.nodes.data(), top.out.data(), top.out.size() * 4);
            for (size_t i = 0; i < pieces.size(); i++) {
                const uint32_t word = (uint32_t)base[i];
                std::memcpy(&hs.nodes[top.deferred[i].at], &word, 4);
            }
            {
                std::atomic<size_t> next_piece{0};
                auto worker = [&]() {
                    for (;;) {
                        size_t i = next_piece.fetch_add(1);
                        if (i >= pieces.size()) break;
                        float *dst = &hs.nodes[base[i] * 32];
                        const std::vector<float> &src = pieces[i].out;
                        std::memcpy(dst, src.data(), src.size() * 4);
                        for (size_t nd = 0; nd < src.size() / 32; nd++)
                            for (int c = 0; c < 4; c++) {
                                uint32_t w;
                                std::memcpy(&w, dst + nd * 32 + 24 + c, 4);
                                if (!(w & RT_CHILD_LEAF)) {  // a node of this piece (empty slots and leaves have the bit set)
                                    w += (uint32_t)base[i];
                                    std::memcpy(dst + nd * 32 + 24 + c, &w, 4);
                                }
                            }
                        std::vector<float>().swap(pieces[i].out);
                    }
                };
                std::vector<std::thread> pool;
                for (unsigned w = 1; w < std::min<size_t>(hw, pieces.size()); w++) pool.emplace_back(worker);
                worker();
                for (auto &t : pool) t.join();
            }
            hs.wide_depth = depth_max;
        } else {
            std::vector<LeafSlot> slots;
            slots.reserve(total);
            collect_leaf_slots(B.nodes, root, rank_of, slots);
            std::vector<uint32_t> order(slots.size());
            for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
            SahBuilder S(slots);
            S.spare_threads = (int)hw - 1;
            int32_t sroot = S.build(order.data(), order.size());
            hs.nodes.reserve((total / 4 + 8) * 32);
            SahCollapser C(S.nodes, slots, hs.nodes);
            C.make(sroot, 1);
            if (C.overflow) {
                err = "internal: leaf encoding overflow";
                return RT_ERR_INTERNAL;
            }
            hs.wide_depth = C.depth_max;
        }
        hs.n_nodes = hs.nodes.size() / 32;
        if (std::getenv("RTCORE_DEBUG_TREE")) {
            double sum = 0.0;
            for (const BNode &nd : B.nodes) sum += half_area(nd.box);
            std::fprintf(stderr, "[rtcore] binary tree: %zu nodes, summed half area %.6g\n", B.nodes.size(), sum);
        }
        if (std::getenv("RTCORE_DEBUG_TREE")) {  // surface-area cost of the wide tree, for comparing builders
            double inner = 0.0, leaf = 0.0;
            size_t n_leaf1 = 0, n_leaf2 = 0, n_inner_slots = 0;
            for (size_t nd = 0; nd < hs.nodes.size() / 32; nd++) {
                const float *f = &hs.nodes[nd * 32];
                uint32_t w[4];
                std::memcpy(w, f + 24, 16);
                for (int c = 0; c < 4; c++) {
                    if (w[c] == RT_CHILD_NONE) continue;
                    double dx = f[12 + c] - f[c], dy = f[16 + c] - f[4 + c], dz = f[20 + c] - f[8 + c];
                    double a = dx * dy + dy * dz + dz * dx;
                    if (w[c] & RT_CHILD_LEAF) { leaf += a; if ((w[c] >> 28) & 7u) n_leaf2++; else n_leaf1++; } else { inner += a; n_inner_slots++; }
                }
            }
            std::fprintf(stderr, "[rtcore] wide tree: %zu nodes, summed half area of node children %.6g, of leaf children %.6g; %zu node slots, %zu single and %zu pair leaf slots\n",
                         hs.nodes.size() / 32, inner, leaf, n_inner_slots, n_leaf1, n_leaf2);
        }
        if (hs.n_nodes >= (1ull << 25)) {  // kernels address nodes with 32-bit byte offsets
            err = "too many tree nodes";
            return RT_ERR_INVALID_ARGUMENT;
        }
        hs.max_stack = 3 * hs.wide_depth + 4;
        for (int k = 0; k < 3; k++) hs.root_lo[k] = B.nodes[root].box.lo[k], hs.root_hi[k] = B.nodes[root].box.hi[k];

        phase("wide nodes");
        // ---- records in rank order
        hs.prims.resize(total * 12);
        hs.shade_records.resize(total * 24);
        parallel_ranges(total, [&](uint64_t p_lo, uint64_t p_hi) {
            for (uint64_t p = p_lo; p < p_hi; p++) {
                uint64_t r = rank_of[p];
                std::memcpy(&hs.prims[r * 12], &rec[p * 12], 48);
                std::memcpy(&hs.shade_records[r * 24], &rec[p * 12], 48);
                std::memcpy(&hs.shade_records[r * 24 + 12], &nrm[p * 12], 48);
            }
        });
    }
    phase("records in rank order");
    hs.n_prims = total;

    return fill_materials_camera_env(d, hs, err);
}

// Materials, camera and environment of a scene description (everything of a HostScene that is not the tree).
int fill_materials_camera_env(const RtSceneDesc *d, HostScene &hs, std::string &err) {
    hs.materials.resize(d->n_materials);
    for (uint64_t i = 0; i < d->n_materials; i++) {
        const RtMaterial &m = d->materials[i];
        if (m.type < RT_MAT_DIELECTRIC || m.type > RT_MAT_GLOSSY) {
            err = "material " + std::to_string(i) + ": unknown type";
            return RT_ERR_INVALID_ARGUMENT;
        }
        RtDevMaterial &o = hs.materials[i];
        std::memset(&o, 0, sizeof(o));
        o.albedo[0] = m.albedo[0], o.albedo[1] = m.albedo[1], o.albedo[2] = m.albedo[2];
        o.type = m.type;
        o.p0 = m.p0, o.p1 = m.p1;
    }
    hs.camera = d->camera;
    hs.env.clear();
    hs.env_w = hs.env_h = 0;
    if (d->env.rgb && d->env.width && d->env.height) {
        hs.env.assign(d->env.rgb, d->env.rgb + (size_t)d->env.width * d->env.height * 3);
        hs.env_w = d->env.width;
        hs.env_h = d->env.height;
    }
    return RT_OK;
}

}  // namespace rtcore
