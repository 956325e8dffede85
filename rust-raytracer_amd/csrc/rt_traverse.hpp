// rt_traverse.hpp -- closest-hit traversal of the 4-wide tree, one ray per lane.
//
// What must equal the reference (src/tracer/bvh.rs:142-161): the primitive it
// returns and that primitive's t.  The reference visits right-then-left and hands
// the left subtree t_max = right hit's t, so among candidates at exactly the same
// distance the outcome depends on the order of visits and on whether a
// primitive's own bound check accepts t == t_max:
//     triangle            t_min <= t <= t_max      accepts a tie  (src/tracer/triangle.rs:64)
//     box, entry face     t_near <= t_max          accepts a tie  (src/tracer/cube.rs:47-51)
//     box, exit face      t_far  <  t_max          rejects a tie  (src/tracer/cube.rs:52)
//     sphere              t_min <  t <  t_max      rejects a tie  (src/tracer/sphere.rs:36,47)
// Primitives are stored in the reference's visit order, so "visited later" is
// "has the larger index".  A candidate c replaces the current best b when
//     c.t < b.t,  or  c.t == b.t and (c.index > b.index ? c accepts ties : !(b accepts ties)).
// With that rule any traversal order gives the reference's answer, and this
// traversal goes front to back and culls with the current best distance.
//
// Box tests use the reference's formula (src/tracer/cube.rs:163-171) verbatim:
// t0 = (min - o) * (1/d), t1 = (max - o) * (1/d), hit iff
// max(max3(min(t0,t1)), t_min) <= min(min3(max(t0,t1)), t_max), with NaN-ignoring
// min/max.  The formula is monotone in the box, so skipping the reference's
// interior boxes that the wide nodes swallowed cannot change which leaves are
// reached; the box that guards each leaf slot is the one the reference tests last
// before that leaf.
#ifndef RT_TRAVERSE_HPP
#define RT_TRAVERSE_HPP

#include "rt_device.hpp"

namespace rtd {

struct Best {
    float t;
    // 2 * visit rank (= index into the primitive array) + 1 if the primitive's own bound check accepts t == t_max;
    // kBestNone = no hit yet.  One register for both: two candidates never share a rank, so comparing keys
    // compares ranks.
    int32_t key;
    RT_DEV int32_t index() const { return key >> 1; }  // -1 = none
};
constexpr int32_t kBestNone = -2;

// `valid` = the primitive's own test passed; written without branches (the lanes of a wave test
// different primitives, so an early exit never lets the wave skip anything)
RT_DEV void consider(Best &b, bool valid, float t, bool ties, int32_t index) {
    const int32_t key = index * 2 + (ties ? 1 : 0);
    const bool tie_goes = (key > b.key) ? ties : ((b.key & 1) == 0);
    const bool closer = valid & ((t < b.t) | ((t == b.t) & tie_goes));
    b.t = closer ? t : b.t;
    b.key = closer ? key : b.key;
}

// ABox::hit without the normal (src/tracer/cube.rs:42-56): candidate t and whether it is the entry face
// `inv` is 1/ray.d per component (ABox::hit computes it itself, src/tracer/cube.rs:43; the value is the same)
RT_DEV bool abox_candidate(F3 lo, F3 hi, const Ray &ray, F3 inv, float t_min, float t_max, float *t_out, bool *entry) {
    F3 t0 = (lo - ray.o) * inv;
    F3 t1 = (hi - ray.o) * inv;
    float t_near = fmaxf(max3(f3(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z))), t_min);
    float t_far = fminf(min3(f3(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z))), t_max);
    if (t_near <= t_far) {
        if (t_near > t_min) {
            *t_out = t_near;
            *entry = true;
            return true;
        }
        if (t_far < t_max) {
            *t_out = t_far;
            *entry = false;
            return true;
        }
    }
    return false;
}
// the face-matching normal of ABox::hit (src/tracer/cube.rs:57-66)
RT_DEV F3 abox_normal(F3 lo, F3 hi, F3 p, bool hollow) {
    F3 n;
    if (fabsf(p.x - hi.x) < kBoxTolerance)
        n = f3(1.0f, 0.0f, 0.0f);
    else if (fabsf(p.y - hi.y) < kBoxTolerance)
        n = f3(0.0f, 1.0f, 0.0f);
    else if (fabsf(p.z - hi.z) < kBoxTolerance)
        n = f3(0.0f, 0.0f, 1.0f);
    else if (fabsf(p.x - lo.x) < kBoxTolerance)
        n = -f3(1.0f, 0.0f, 0.0f);
    else if (fabsf(p.y - lo.y) < kBoxTolerance)
        n = -f3(0.0f, 1.0f, 0.0f);
    else if (fabsf(p.z - lo.z) < kBoxTolerance)
        n = -f3(0.0f, 0.0f, 1.0f);
    else
        n = f3(0.0f, 0.0f, 0.0f);
    return n * (hollow ? -1.0f : 1.0f);
}
// the local ray of Cube::hit (src/tracer/cube.rs:107-111)
RT_DEV Ray cube_local_ray(F3 center, Rotor rot, const Ray &ray) {
    Rotor rev = reversed(rot);
    Ray l;
    l.o = rotate(rev, ray.o - center) + center;
    l.d = normalized(rotate(rev, ray.d));
    return l;
}

// Triangle::hit (src/tracer/triangle.rs:37-66) on the record (r0, r1, r2); e1, e2 are stored, they are the same floats.
// The reference's early returns are one predicate over the same expressions; `valid` = there is a triangle to test.
RT_DEV void test_triangle(float4 r0, float4 r1, float4 r2, int32_t index, const Ray &ray, float t_min, float t_max,
                          bool valid, Best &best) {
    const bool two_sided = (__float_as_uint(r2.z) & RT_PRIM_FLAG) != 0;
    F3 v0 = f3(r0.x, r0.y, r0.z), e1 = f3(r0.w, r1.x, r1.y), e2 = f3(r1.z, r1.w, r2.x);
    F3 h = cross(ray.d, e2);
    float a = dot(e1, h);
    bool ok = valid & !((!two_sided & ((__float_as_uint(a) >> 31) != 0u)) | (fabsf(a) < 1e-6f));
    float f = 1.0f / a;
    F3 s = ray.o - v0;
    float u = f * dot(s, h);
    ok = ok & ((u >= 0.0f) & (u <= 1.0f));
    F3 q = cross(s, e1);
    float v = f * dot(ray.d, q);
    ok = ok & !((v < 0.0f) | (u + v > 1.0f));
    float t = f * dot(e2, q);
    ok = ok & !((t < t_min) | (t > t_max));
    consider(best, ok, t, true, index);
}

// Tests the primitive whose record is (r0, r1, r2), of a type other than triangle, and offers its candidate to `best`.
// t_max is the ray's own bound.
RT_DEV void test_record_other(float4 r0, float4 r1, float4 r2, int32_t index, const Ray &ray, F3 inv, float t_min,
                              float t_max, Best &best) {
    uint32_t meta = __float_as_uint(r2.z);
    uint32_t type = meta & RT_PRIM_TYPE_MASK;
    if (type == RT_PRIM_SPHERE) {
        // Sphere::hit (src/tracer/sphere.rs:28-58)
        F3 c = f3(r0.x, r0.y, r0.z);
        float radius = r0.w;
        F3 oc = ray.o - c;
        float half_b = dot(oc, ray.d);
        float disc = half_b * half_b - (mag_sq(oc) - (radius * radius));
        if (disc > 0.0f) {
            float h = sqrtf(disc);
            float temp = -half_b - h;
            if (temp < t_max && temp > t_min) {
                consider(best, true, temp, false, index);
                return;
            }
            temp = -half_b + h;
            if (temp < t_max && temp > t_min) consider(best, true, temp, false, index);
        }
    } else if (type == RT_PRIM_ABOX) {
        F3 lo = f3(r0.x, r0.y, r0.z), hi = f3(r0.w, r1.x, r1.y);
        float t;
        bool entry;
        if (abox_candidate(lo, hi, ray, inv, t_min, t_max, &t, &entry)) consider(best, true, t, entry, index);
    } else {
        F3 center = f3(r0.x, r0.y, r0.z), half = f3(r0.w, r1.x, r1.y);
        Rotor rot{r1.z, r1.w, r2.x, r2.y};
        Ray local = cube_local_ray(center, rot, ray);
        float t;
        bool entry;
        F3 linv = f3(1.0f / local.d.x, 1.0f / local.d.y, 1.0f / local.d.z);
        if (abox_candidate(center - half, center + half, local, linv, t_min, t_max, &t, &entry))
            consider(best, true, t, entry, index);
    }
}

// Tests the primitive whose record is (r0, r1, r2), of any type.
RT_DEV void test_record(float4 r0, float4 r1, float4 r2, int32_t index, const Ray &ray, F3 inv, float t_min,
                        float t_max, Best &best) {
    if ((__float_as_uint(r2.z) & RT_PRIM_TYPE_MASK) == RT_PRIM_TRIANGLE)
        test_triangle(r0, r1, r2, index, ray, t_min, t_max, true, best);
    else
        test_record_other(r0, r1, r2, index, ray, inv, t_min, t_max, best);
}

// HitRecord fields of the winning primitive (visit rank `index`, distance `t`): normal, material, primitive id.
RT_DEV void finish_hit(const RtSceneView &sc, const Ray &ray, float t, int32_t index, F3 *normal, uint32_t *material,
                       int32_t *prim_id) {
    const float4 *r = (const float4 *)sc.shade_records + (size_t)index * RT_SHADE_RECORD_FLOAT4S;
    float4 r0 = r[0], r1 = r[1], r2 = r[2];
    uint32_t meta = __float_as_uint(r2.z);
    uint32_t type = meta & RT_PRIM_TYPE_MASK;
    bool flag = (meta & RT_PRIM_FLAG) != 0;
    *material = meta >> RT_PRIM_MAT_SHIFT;
    *prim_id = (int32_t)__float_as_uint(r2.w);
    if (type == RT_PRIM_TRIANGLE) {
        // The traversal carries only (t, rank): the barycentrics of the winning triangle are worked
        // out again here, by the same operations on the same floats as in test_record, so they
        // are the same bits (src/tracer/triangle.rs:41-58).
        F3 v0 = f3(r0.x, r0.y, r0.z), e1 = f3(r0.w, r1.x, r1.y), e2 = f3(r1.z, r1.w, r2.x);
        F3 h = cross(ray.d, e2);
        float f = 1.0f / dot(e1, h);
        F3 s = ray.o - v0;
        float u = f * dot(s, h);
        float v = f * dot(ray.d, cross(s, e1));
        // src/tracer/triangle.rs:68-70
        const float4 *nn = r + RT_PRIM_FLOAT4S;
        float4 a = nn[0], b = nn[1], c = nn[2];
        F3 n0 = f3(a.x, a.y, a.z), n1 = f3(b.x, b.y, b.z), n2 = f3(c.x, c.y, c.z);
        *normal = normalized((1.0f - (u + v)) * n0 + u * n1 + v * n2);
    } else if (type == RT_PRIM_SPHERE) {
        F3 p = ray_at(ray, t);
        *normal = normalized(p - f3(r0.x, r0.y, r0.z));
    } else if (type == RT_PRIM_ABOX) {
        *normal = abox_normal(f3(r0.x, r0.y, r0.z), f3(r0.w, r1.x, r1.y), ray_at(ray, t), flag);
    } else {
        // Cube::hit (src/tracer/cube.rs:110-117): the face is matched in the box's frame
        F3 center = f3(r0.x, r0.y, r0.z), half = f3(r0.w, r1.x, r1.y);
        Rotor rot{r1.z, r1.w, r2.x, r2.y};
        Ray local = cube_local_ray(center, rot, ray);
        F3 nl = abox_normal(center - half, center + half, ray_at(local, t), flag);
        *normal = rotate(rot, nl);
    }
}

// Per-lane traversal stack of (child word, entry distance) pairs: the first RT_LDS_STACK
// entries of a lane live in LDS (entry-major, 8-byte entries, so the lanes of a wave hit
// consecutive banks), deeper ones in an HBM area with the same layout over all lanes of the grid.
struct Stack {
    uint2 *lds;  // [RT_LDS_STACK][block threads], this thread's column
    uint32_t lds_stride;
    uint2 *ovf;  // [max_stack - RT_LDS_STACK][grid threads]; the thread's column is worked out on use (rare)
    uint32_t ovf_stride;
    int sp;
    RT_DEV uint2 *deep(int slot) const {
        return ovf + ((size_t)(slot - RT_LDS_STACK) * ovf_stride + (blockIdx.x * blockDim.x + threadIdx.x));
    }
    RT_DEV void store(int slot, uint32_t w, float t) {
        if (slot < RT_LDS_STACK)
            lds[slot * lds_stride] = make_uint2(w, __float_as_uint(t));
        else
            *deep(slot) = make_uint2(w, __float_as_uint(t));
    }
    RT_DEV void pop(uint32_t *w, float *t) {
        sp--;
        uint2 e = (sp < RT_LDS_STACK) ? lds[sp * lds_stride] : *deep(sp);
        *w = e.x;
        *t = __uint_as_float(e.y);
    }
};

// Bvh::hit for one ray as a resumable state machine, so that a wave can keep its lanes
// busy with different rays: begin() loads a ray and tests the root box, step() handles one
// node or one leaf and picks the next entry, and reports when the stack has run empty.
struct Traversal {
    Ray ray;
    F3 inv;
    Best best;
    uint32_t entry;
};

// Returns false when the result is already known to be a miss (no traversal needed).
template <bool COUNT>
RT_DEV bool traversal_begin(const RtSceneView &sc, Traversal &tr, Stack &st, float t_min, float t_max,
                            uint32_t *visits) {
    tr.best.t = t_max;
    tr.best.key = kBestNone;
    st.sp = 0;
    tr.entry = 0;  // node 0
    const Ray &ray = tr.ray;
    // A ray whose direction (or origin) is NaN in every component passes every box test of the
    // reference (NaN-ignoring min/max leave t_min <= t_max) and fails every primitive test, so it
    // would walk the whole tree only to miss.  Ray::new turns a zero or NaN direction into such a
    // ray.  The result is known without the walk.
    // The same holds when every axis has a NaN in its origin or its direction component: every box test ignores
    // every axis, and every primitive test mixes the components and fails.
    if ((ray.d.x != ray.d.x || ray.o.x != ray.o.x) && (ray.d.y != ray.d.y || ray.o.y != ray.o.y) &&
        (ray.d.z != ray.d.z || ray.o.z != ray.o.z))
        return false;
    tr.inv = f3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    // the root's own box (src/tracer/bvh.rs:144-147)
    F3 t0 = (f3(sc.root_lo[0], sc.root_lo[1], sc.root_lo[2]) - ray.o) * tr.inv;
    F3 t1 = (f3(sc.root_hi[0], sc.root_hi[1], sc.root_hi[2]) - ray.o) * tr.inv;
    float tn = fmaxf(max3(f3(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z))), t_min);
    float tf = fminf(min3(f3(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z))), t_max);
    if (COUNT) (*visits)++;
    return tn <= tf;
}

// What a lane of the closest-hit kernel is doing.  The two kinds of step come first, so that the leaf bit of the
// child word a lane goes on with IS its next state; a lane with state >= kParked has no ray at work.
constexpr uint32_t kAtNode = 0u, kAtLeaf = 1u, kPopping = 2u, kParked = 3u, kEmpty = 4u;
static_assert((RT_CHILD_LEAF >> 31) == kAtLeaf, "a child word's top bit is the lane's next state");

// One turn of taking the next entry from the stack: the two top entries are read in one LDS round trip; an entry
// whose box starts beyond the best hit is dropped.  Returns the lane's next state: at a node / at a leaf (tr.entry
// is the entry taken), still popping (both entries were dead and more remain), or parked (the stack has run empty:
// the traversal is finished).  Straight-line code but for the rare lane whose top entry is in the overflow area.
RT_DEV uint32_t traversal_pop_turn(Traversal &tr, Stack &st) {
    if (st.sp > RT_LDS_STACK) {
        uint32_t w;
        float t;
        st.pop(&w, &t);
        if (t <= tr.best.t) {
            tr.entry = w;
            return w >> 31;
        }
        return kPopping;
    }
    const int top = st.sp >= 1 ? st.sp - 1 : 0, below = st.sp >= 2 ? st.sp - 2 : 0;
    uint2 e1 = st.lds[top * st.lds_stride], e2 = st.lds[below * st.lds_stride];
    // (both entries are read whole and unconditionally -- row 0 exists for every lane -- so that the turn is two LDS
    // reads and a handful of selects, no branch)
    asm volatile("" : "+v"(e1.x), "+v"(e1.y), "+v"(e2.x), "+v"(e2.y));
    const bool ok1 = (st.sp >= 1) & (__uint_as_float(e1.y) <= tr.best.t);
    const bool ok2 = (st.sp >= 2) & (__uint_as_float(e2.y) <= tr.best.t);
    const uint32_t w = ok1 ? e1.x : e2.x;
    st.sp = ok1 ? top : below;
    tr.entry = w;
    const uint32_t dry = st.sp == 0 ? kParked : kPopping;
    return (ok1 | ok2) ? (w >> 31) : dry;
}

// tr.entry is a leaf: tests its primitives.  The lane then wants the next stack entry.
template <bool COUNT>
RT_DEV void traversal_leaf_step(const RtSceneView &sc, Traversal &tr, float t_min, float t_max, uint32_t *tests) {
    const float4 *__restrict__ prims = (const float4 *)sc.prims;
    const uint32_t entry = tr.entry;
    int32_t first = (int32_t)(entry & RT_LEAF_FIRST_MASK);
    int32_t count = (int32_t)((entry >> 28) & 7u) + 1;
    // A leaf holds one or two primitives (more only if a builder packs them): both records are
    // fetched before the first test so that their latencies overlap.
    const float4 *r = prims + (size_t)first * RT_PRIM_FLOAT4S;
    const bool two = count > 1;
    float4 a0 = r[0], a1 = r[1], a2 = r[2];
    const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // (not a copy of the first record: that would wait for it)
    float4 b0 = zero, b1 = zero, b2 = zero;
    if (two) b0 = r[3], b1 = r[4], b2 = r[5];
    if (COUNT) (*tests) += (uint32_t)count;
    test_record(a0, a1, a2, first, tr.ray, tr.inv, t_min, t_max, tr.best);
    if (two) test_record(b0, b1, b2, first + 1, tr.ray, tr.inv, t_min, t_max, tr.best);
    for (int32_t k = 2; k < count; k++) {
        const float4 *q = prims + (size_t)(first + k) * RT_PRIM_FLOAT4S;
        test_record(q[0], q[1], q[2], first + k, tr.ray, tr.inv, t_min, t_max, tr.best);
    }
}

// The four child boxes of a wide node against a ray: the slab test of src/tracer/cube.rs:163-171 per child, then the
// hit children sorted by entry distance.  In: the node's seven 16-byte words.  Out: key[0..nh) ascending entry
// distances with their child words in cw[0..nh); returns nh.
RT_DEV int node_children(const float4 nx4, const float4 ny4, const float4 nz4, const float4 fx4, const float4 fy4,
                         const float4 fz4, const float4 cwf, const F3 o, const F3 inv, float t_min, float t_best,
                         float key[4], uint32_t cw[4]) {
    cw[0] = __float_as_uint(cwf.x), cw[1] = __float_as_uint(cwf.y), cw[2] = __float_as_uint(cwf.z), cw[3] = __float_as_uint(cwf.w);
    // the reference's t0 = (min - o) * inv, t1 = (max - o) * inv (src/tracer/cube.rs:164-165) for two
    // children at a time: gfx950 subtracts and multiplies two floats per instruction
    // (v_pk_add_f32 / v_pk_mul_f32, IEEE results, no contraction), min / max have no packed form.
    typedef float Pair __attribute__((ext_vector_type(2)));
    const Pair ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    const Pair ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    const Pair t0x[2] = {(Pair{nx4.x, nx4.y} - ox) * ix, (Pair{nx4.z, nx4.w} - ox) * ix};
    const Pair t1x[2] = {(Pair{fx4.x, fx4.y} - ox) * ix, (Pair{fx4.z, fx4.w} - ox) * ix};
    const Pair t0y[2] = {(Pair{ny4.x, ny4.y} - oy) * iy, (Pair{ny4.z, ny4.w} - oy) * iy};
    const Pair t1y[2] = {(Pair{fy4.x, fy4.y} - oy) * iy, (Pair{fy4.z, fy4.w} - oy) * iy};
    const Pair t0z[2] = {(Pair{nz4.x, nz4.y} - oz) * iz, (Pair{nz4.z, nz4.w} - oz) * iz};
    const Pair t1z[2] = {(Pair{fz4.x, fz4.y} - oz) * iz, (Pair{fz4.z, fz4.w} - oz) * iz};
    const float kInf = __int_as_float(0x7f800000);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        // per-axis min / max of the two plane distances and the interval test (src/tracer/cube.rs:166-170).
        const float ax0 = t0x[c >> 1][c & 1], ax1 = t1x[c >> 1][c & 1];
        const float ay0 = t0y[c >> 1][c & 1], ay1 = t1y[c >> 1][c & 1];
        const float az0 = t0z[c >> 1][c & 1], az1 = t1z[c >> 1][c & 1];
        float nxv = fminf(ax0, ax1), fxv = fmaxf(ax0, ax1);
        float nyv = fminf(ay0, ay1), fyv = fmaxf(ay0, ay1);
        float nzv = fminf(az0, az1), fzv = fmaxf(az0, az1);
        float tn = fmaxf(fmaxf(fmaxf(nxv, nyv), nzv), t_min);
        float tf = fminf(fminf(fminf(fxv, fyv), fzv), t_best);
        bool hit = (cw[c] != RT_CHILD_NONE) && (tn <= tf);
        key[c] = hit ? tn : kInf;
    }
    // sort the four (key, word) pairs ascending: (0,1) (2,3) (0,2) (1,3) (1,2)
#define RT_CSWAP(i, j)                    \
    if (key[j] < key[i]) {                \
        float tk = key[i];                \
        key[i] = key[j];                  \
        key[j] = tk;                      \
        uint32_t tw = cw[i];              \
        cw[i] = cw[j];                    \
        cw[j] = tw;                       \
    }
    RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
    return (int)(key[0] < kInf) + (int)(key[1] < kInf) + (int)(key[2] < kInf) + (int)(key[3] < kInf);
}

// The same slab tests for the lane-bound kernel's node step.  Out: the four (entry distance, child word) pairs sorted
// ascending with the children that are NOT hit first (their key is -inf), so the hit children are the last nh pairs,
// nearest first among them.  Straight-line code: the lanes of a wave hit different children, so a branch per
// exchange would be taken by every wave anyway and only adds the moves of both sides.
// An empty slot needs no test of its own: the builders give it a box at +infinity, which no ray that gets this far
// hits (every axis of such a ray is either ignored -- a NaN in its origin or direction -- or gives +-inf on both
// planes, and traversal_begin answers the rays whose three axes are all ignored).
RT_DEV int node_children_sorted(const float4 nx4, const float4 ny4, const float4 nz4, const float4 fx4, const float4 fy4,
                                const float4 fz4, const float4 cwf, const F3 o, const F3 inv, float t_min, float t_best,
                                float key[4], uint32_t cw[4]) {
    cw[0] = __float_as_uint(cwf.x), cw[1] = __float_as_uint(cwf.y), cw[2] = __float_as_uint(cwf.z), cw[3] = __float_as_uint(cwf.w);
    typedef float Pair __attribute__((ext_vector_type(2)));
    const Pair ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    const Pair ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    const Pair t0x[2] = {(Pair{nx4.x, nx4.y} - ox) * ix, (Pair{nx4.z, nx4.w} - ox) * ix};
    const Pair t1x[2] = {(Pair{fx4.x, fx4.y} - ox) * ix, (Pair{fx4.z, fx4.w} - ox) * ix};
    const Pair t0y[2] = {(Pair{ny4.x, ny4.y} - oy) * iy, (Pair{ny4.z, ny4.w} - oy) * iy};
    const Pair t1y[2] = {(Pair{fy4.x, fy4.y} - oy) * iy, (Pair{fy4.z, fy4.w} - oy) * iy};
    const Pair t0z[2] = {(Pair{nz4.x, nz4.y} - oz) * iz, (Pair{nz4.z, nz4.w} - oz) * iz};
    const Pair t1z[2] = {(Pair{fz4.x, fz4.y} - oz) * iz, (Pair{fz4.z, fz4.w} - oz) * iz};
    const float kNegInf = __int_as_float(0xff800000);
    int nh = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float ax0 = t0x[c >> 1][c & 1], ax1 = t1x[c >> 1][c & 1];
        const float ay0 = t0y[c >> 1][c & 1], ay1 = t1y[c >> 1][c & 1];
        const float az0 = t0z[c >> 1][c & 1], az1 = t1z[c >> 1][c & 1];
        float nxv = fminf(ax0, ax1), fxv = fmaxf(ax0, ax1);
        float nyv = fminf(ay0, ay1), fyv = fmaxf(ay0, ay1);
        float nzv = fminf(az0, az1), fzv = fmaxf(az0, az1);
        float tn = fmaxf(fmaxf(fmaxf(nxv, nyv), nzv), t_min);
        float tf = fminf(fminf(fminf(fxv, fyv), fzv), t_best);
        const bool hit = tn <= tf;
        key[c] = hit ? tn : kNegInf;
        nh += hit ? 1 : 0;
    }
#define RT_CSEL(i, j)                                   \
    {                                                   \
        const bool sw = key[j] < key[i];                \
        const float ki = sw ? key[j] : key[i];          \
        const float kj = sw ? key[i] : key[j];          \
        const uint32_t wi = sw ? cw[j] : cw[i];         \
        const uint32_t wj = sw ? cw[i] : cw[j];         \
        key[i] = ki, key[j] = kj, cw[i] = wi, cw[j] = wj; \
    }
    RT_CSEL(0, 1) RT_CSEL(2, 3) RT_CSEL(0, 2) RT_CSEL(1, 3) RT_CSEL(1, 2)
#undef RT_CSEL
    return nh;
}

// The seven 16-byte words of a wide node.
struct NodeWords {
    float4 nx4, ny4, nz4, fx4, fy4, fz4, cwf;
};

// A lane fetches its own node: seven 16-byte loads from one 128-byte line.
RT_DEV void node_fetch_own(const RtSceneView &sc, uint32_t entry, NodeWords &w) {
    const char *__restrict__ nodes = (const char *)sc.nodes;
    const uint32_t base = entry * (RT_NODE_FLOAT4S * 16u);  // 32-bit byte offsets from the (uniform) node array base
    w.nx4 = *(const float4 *)(nodes + (base + 0u));
    w.ny4 = *(const float4 *)(nodes + (base + 16u));
    w.nz4 = *(const float4 *)(nodes + (base + 32u));
    w.fx4 = *(const float4 *)(nodes + (base + 48u));
    w.fy4 = *(const float4 *)(nodes + (base + 64u));
    w.fz4 = *(const float4 *)(nodes + (base + 80u));
    w.cwf = *(const float4 *)(nodes + (base + 96u));
}

// After the slab tests of a node step: pushes the far hit children and continues with the nearest.  `key` / `cw` are the
// sorted pairs (the hit children are the pairs [4 - nh, 4), nearest first).  Returns the lane's next state: at the
// nearest hit child (a node or a leaf), or popping when no child was hit.
RT_DEV uint32_t traversal_descend(const int nh, const float key[4], const uint32_t cw[4], Traversal &tr, Stack &st) {
    // the nearest hit child is pair 4 - nh: three selects (kept apart from the stores' branch below, so that they stay
    // selects)
    uint32_t next = cw[3];
    next = nh >= 2 ? cw[2] : next;
    next = nh >= 3 ? cw[1] : next;
    next = nh >= 4 ? cw[0] : next;
    uint32_t after = nh == 0 ? kPopping : (next >> 31);
    asm volatile("" : "+v"(next), "+v"(after));
    // The farthest goes to the stack first, the nearest continues.  Pairs 3, 2, 1 are written to the three entries above
    // the top whether they are hit children or not: what is not one lies above the new top, where nothing reads it.
    if (st.sp + 3 <= RT_LDS_STACK) {
        uint2 *top = st.lds + st.sp * st.lds_stride;
        top[0] = make_uint2(cw[3], __float_as_uint(key[3]));
        top[st.lds_stride] = make_uint2(cw[2], __float_as_uint(key[2]));
        top[2 * st.lds_stride] = make_uint2(cw[1], __float_as_uint(key[1]));
    } else {
        if (nh > 1) st.store(st.sp, cw[3], key[3]);
        if (nh > 2) st.store(st.sp + 1, cw[2], key[2]);
        if (nh > 3) st.store(st.sp + 2, cw[1], key[1]);
    }
    st.sp += nh > 1 ? nh - 1 : 0;
    tr.entry = next;
    return after;
}

// tr.entry is a node whose words are `w`: tests its four child boxes, pushes the far hits and continues with the
// nearest.  Returns the lane's next state.
template <bool COUNT>
RT_DEV uint32_t traversal_node_process(const NodeWords &w, Traversal &tr, Stack &st, float t_min, uint32_t *visits) {
    if (COUNT) (*visits)++;
    float key[4];
    uint32_t cw[4];
    const int nh = node_children_sorted(w.nx4, w.ny4, w.nz4, w.fx4, w.fy4, w.fz4, w.cwf, tr.ray.o, tr.inv, t_min, tr.best.t, key, cw);
    return traversal_descend(nh, key, cw, tr, st);
}

template <bool COUNT>
RT_DEV uint32_t traversal_node_step(const RtSceneView &sc, Traversal &tr, Stack &st, float t_min, uint32_t *visits) {
    NodeWords w;
    node_fetch_own(sc, tr.entry, w);
    return traversal_node_process<COUNT>(w, tr, st, t_min, visits);
}

// ---- the same walk over compact nodes (rt_types.h) -----------------------------------------------------------
// A compact child box contains the exact one, and the slab test is monotone in the box, so a ray that the reference's
// test lets into a leaf's guard box gets through every compact box above it PROVIDED the arithmetic here never rejects
// what exact arithmetic on the compact box would accept.  The distance to plane k is evaluated as k * A + B with
// A = step / d and B = (origin - o) / d per axis; against the reference's two roundings on the guard box and the
// roundings here, the near distances are lowered and the far distances raised by kGridSlack * (|B| + 255 |A|), which
// bounds every plane distance of the node on that axis (2^-20 of it: sixteen roundings' worth).  An axis whose A or B
// is not finite (d = 0, overflow) yields NaN or -inf / +inf there and is ignored by the NaN-ignoring min / max, which
// again only lets more rays through.  Entry distances of compact boxes are lower bounds: they order the children and
// drop stack entries beyond the best hit, nothing else.
constexpr float kGridSlack = 9.5367431640625e-07f;  // 2^-20

struct QNodeWords {
    float4 a, b, c, cwf;
};

RT_DEV void qnode_fetch_own(const RtSceneView &sc, uint32_t entry, QNodeWords &w) {
    const char *__restrict__ nodes = (const char *)sc.qnodes;
    const uint32_t base = entry * (RT_QNODE_FLOAT4S * 16u);
    w.a = *(const float4 *)(nodes + (base + 0u));
    w.b = *(const float4 *)(nodes + (base + 16u));
    w.c = *(const float4 *)(nodes + (base + 32u));
    w.cwf = *(const float4 *)(nodes + (base + 48u));
}

template <int K>
RT_DEV float grid_byte(uint32_t v) {
    return (float)((v >> (8 * K)) & 0xffu);  // v_cvt_f32_ubyteK
}

struct GridAxis {
    float a, b_near, b_far, bound;
    uint32_t q_near, q_far;
};
RT_DEV GridAxis grid_axis(float origin, float step, uint32_t q_lo, uint32_t q_hi, float o, float inv) {
    GridAxis g;
    g.a = step * inv;
    const float b = (origin - o) * inv;
    g.bound = fmaf(fabsf(g.a), 255.0f, fabsf(b));
    g.b_near = fmaf(-kGridSlack, g.bound, b);
    g.b_far = fmaf(kGridSlack, g.bound, b);
    const bool neg = inv < 0.0f;  // the planes swap roles for a ray that runs down the axis
    g.q_near = neg ? q_hi : q_lo;
    g.q_far = neg ? q_lo : q_hi;
    return g;
}

template <int C>
RT_DEV bool grid_child(const GridAxis &x, const GridAxis &y, const GridAxis &z, uint32_t word, float t_min, float t_best,
                       float *key) {
    const float tn = fmaxf(fmaxf(fmaxf(fmaf(grid_byte<C>(x.q_near), x.a, x.b_near), fmaf(grid_byte<C>(y.q_near), y.a, y.b_near)),
                                 fmaf(grid_byte<C>(z.q_near), z.a, z.b_near)),
                           t_min);
    const float tf = fminf(fminf(fminf(fmaf(grid_byte<C>(x.q_far), x.a, x.b_far), fmaf(grid_byte<C>(y.q_far), y.a, y.b_far)),
                                 fmaf(grid_byte<C>(z.q_far), z.a, z.b_far)),
                           t_best);
    const bool hit = (tn <= tf) && (word != RT_CHILD_NONE);
    *key = hit ? tn : __int_as_float(0xff800000);
    return hit;
}

// The rare ray for which A or B is not finite on some axis (it runs parallel to an axis: 1 / d is infinite; or
// something overflowed; or the node has an open axis): every plane is worked out, origin + k * step in one rounding --
// a low plane still at or below the exact box's, a high plane at or above -- and goes through the reference's formula,
// which handles those cases the way the reference does (a parallel ray outside the slab is turned away; ignoring the
// axis instead would send a ray that runs along an axis through every node its line of sight never leaves).
template <int C>
RT_DEV bool plane_child(const QNodeWords &w, uint32_t word, const F3 o, const F3 inv, float t_min, float t_best, float *key) {
    const F3 lo = f3(fmaf(grid_byte<C>(__float_as_uint(w.b.z)), w.a.w, w.a.x), fmaf(grid_byte<C>(__float_as_uint(w.b.w)), w.b.x, w.a.y),
                     fmaf(grid_byte<C>(__float_as_uint(w.c.x)), w.b.y, w.a.z));
    const F3 hi = f3(fmaf(grid_byte<C>(__float_as_uint(w.c.y)), w.a.w, w.a.x), fmaf(grid_byte<C>(__float_as_uint(w.c.z)), w.b.x, w.a.y),
                     fmaf(grid_byte<C>(__float_as_uint(w.c.w)), w.b.y, w.a.z));
    const F3 t0 = (lo - o) * inv, t1 = (hi - o) * inv;
    const float tn = fmaxf(max3(f3(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z))), t_min);
    const float tf = fminf(min3(f3(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z))), t_best);
    const bool hit = (tn <= tf) && (word != RT_CHILD_NONE);
    *key = hit ? tn : __int_as_float(0xff800000);
    return hit;
}

RT_DEV int qnode_children_sorted(const QNodeWords &w, const F3 o, const F3 inv, float t_min, float t_best, float key[4],
                                 uint32_t cw[4]) {
    cw[0] = __float_as_uint(w.cwf.x), cw[1] = __float_as_uint(w.cwf.y), cw[2] = __float_as_uint(w.cwf.z), cw[3] = __float_as_uint(w.cwf.w);
    const GridAxis x = grid_axis(w.a.x, w.a.w, __float_as_uint(w.b.z), __float_as_uint(w.c.y), o.x, inv.x);
    const GridAxis y = grid_axis(w.a.y, w.b.x, __float_as_uint(w.b.w), __float_as_uint(w.c.z), o.y, inv.y);
    const GridAxis z = grid_axis(w.a.z, w.b.y, __float_as_uint(w.c.x), __float_as_uint(w.c.w), o.z, inv.z);
    int nh = 0;
    nh += grid_child<0>(x, y, z, cw[0], t_min, t_best, &key[0]) ? 1 : 0;
    nh += grid_child<1>(x, y, z, cw[1], t_min, t_best, &key[1]) ? 1 : 0;
    nh += grid_child<2>(x, y, z, cw[2], t_min, t_best, &key[2]) ? 1 : 0;
    nh += grid_child<3>(x, y, z, cw[3], t_min, t_best, &key[3]) ? 1 : 0;
    // any bound that is infinite or not a number makes the sum so (the bounds are sums of magnitudes: no cancellation)
    const float kHuge = 3.0e38f;
    if (!((x.bound + y.bound) + z.bound < kHuge)) {
        nh = plane_child<0>(w, cw[0], o, inv, t_min, t_best, &key[0]) ? 1 : 0;
        nh += plane_child<1>(w, cw[1], o, inv, t_min, t_best, &key[1]) ? 1 : 0;
        nh += plane_child<2>(w, cw[2], o, inv, t_min, t_best, &key[2]) ? 1 : 0;
        nh += plane_child<3>(w, cw[3], o, inv, t_min, t_best, &key[3]) ? 1 : 0;
    }
#define RT_CSEL(i, j)                                   \
    {                                                   \
        const bool sw = key[j] < key[i];                \
        const float ki = sw ? key[j] : key[i];          \
        const float kj = sw ? key[i] : key[j];          \
        const uint32_t wi = sw ? cw[j] : cw[i];         \
        const uint32_t wj = sw ? cw[i] : cw[j];         \
        key[i] = ki, key[j] = kj, cw[i] = wi, cw[j] = wj; \
    }
    // (all five exchanges are needed: descend relies on the hit children being the LAST pairs, which a partly
    // sorted quadruple does not give -- the pair it would go on with could be an empty slot's)
    RT_CSEL(0, 1) RT_CSEL(2, 3) RT_CSEL(0, 2) RT_CSEL(1, 3) RT_CSEL(1, 2)
#undef RT_CSEL
    return nh;
}

template <bool COUNT>
RT_DEV uint32_t traversal_qnode_step(const RtSceneView &sc, Traversal &tr, Stack &st, float t_min, uint32_t *visits) {
    QNodeWords w;
    qnode_fetch_own(sc, tr.entry, w);
    if (COUNT) (*visits)++;
    float key[4];
    uint32_t cw[4];
    const int nh = qnode_children_sorted(w, tr.ray.o, tr.inv, t_min, tr.best.t, key, cw);
    return traversal_descend(nh, key, cw, tr, st);
}

// tr.entry is a leaf reached through compact boxes: its guard box -- the box the reference tests last before these
// primitives -- is tested first, by the reference's formula, and the primitives count only if the ray gets through it.
// Guard and records come from the leaf's block, one 128-byte line.
template <bool COUNT>
RT_DEV void traversal_guarded_leaf_step(const RtSceneView &sc, Traversal &tr, float t_min, float t_max, uint32_t *tests) {
    const uint32_t entry = tr.entry;
    const int32_t count = (int32_t)((entry >> 28) & 7u) + 1;
    const float4 *__restrict__ b = (const float4 *)sc.leaves + (size_t)(entry & RT_LEAF_FIRST_MASK) * RT_LEAF_BLOCK_FLOAT4S;
    const bool two = count > 1;
    const float4 g0 = b[0], g1 = b[1];
    float4 a0 = b[2], a1 = b[3], a2 = b[4];
    const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // (not a copy of the first record: that would wait for it)
    float4 b0 = zero, b1 = zero, b2 = zero;
    if (two) b0 = b[5], b1 = b[6], b2 = b[7];
    // src/tracer/cube.rs:163-171 on the guard box
    const F3 t0 = (f3(g0.x, g0.y, g0.z) - tr.ray.o) * tr.inv;
    const F3 t1 = (f3(g0.w, g1.x, g1.y) - tr.ray.o) * tr.inv;
    const float tn = fmaxf(max3(f3(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z))), t_min);
    const float tf = fminf(min3(f3(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z))), tr.best.t);
    // (decided here, before the primitives' tests, and kept as one register: the guard's words are then free)
    uint32_t open_bit = tn <= tf ? 1u : 0u;
    int32_t first = (int32_t)__float_as_uint(g1.z);
    asm volatile("" : "+v"(open_bit), "+v"(first));
    const bool open = open_bit != 0u;
    if (COUNT) (*tests) += open ? (uint32_t)count : 0u;
    Best best = tr.best;
    // Meshes are what large scenes are made of, so the triangle test is straight-line code that every lane of the
    // step runs twice (under "is a triangle" / "is there and is a triangle": the record that is not there reads as
    // zeros), and the other types' tests sit behind one branch that a wave of mesh leaves never takes.  The order in
    // which a leaf's candidates are offered does not matter (see `consider`).
    const bool tri_a = (__float_as_uint(a2.z) & RT_PRIM_TYPE_MASK) == RT_PRIM_TRIANGLE;
    const bool tri_b = (__float_as_uint(b2.z) & RT_PRIM_TYPE_MASK) == RT_PRIM_TRIANGLE;
    if (__builtin_amdgcn_ballot_w64(!tri_a | (two & !tri_b) | (count > 2)) != 0) {
        if (!tri_a) test_record_other(a0, a1, a2, first, tr.ray, tr.inv, t_min, t_max, best);
        if (two & !tri_b) test_record_other(b0, b1, b2, first + 1, tr.ray, tr.inv, t_min, t_max, best);
        for (int32_t k = 2; k < count; k += 2) {  // (no builder makes such leaves today)
            const float4 *q = b + (size_t)(k >> 1) * RT_LEAF_BLOCK_FLOAT4S;
            test_record(q[2], q[3], q[4], first + k, tr.ray, tr.inv, t_min, t_max, best);
            if (k + 1 < count) test_record(q[5], q[6], q[7], first + k + 1, tr.ray, tr.inv, t_min, t_max, best);
        }
    }
    // (and a wave whose leaves hold no triangle at all -- a scene of spheres and boxes -- skips the triangle code)
    if (__builtin_amdgcn_ballot_w64(tri_a | (two & tri_b)) != 0) {
        test_triangle(a0, a1, a2, first, tr.ray, t_min, t_max, tri_a, best);
        test_triangle(b0, b1, b2, first + 1, tr.ray, t_min, t_max, two & tri_b, best);
    }
    tr.best.t = open ? best.t : tr.best.t;
    tr.best.key = open ? best.key : tr.best.key;
}

}  // namespace rtd
#endif
