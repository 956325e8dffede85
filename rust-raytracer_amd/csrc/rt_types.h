// rt_types.h -- data layout shared by the host-side scene builder and the HIP kernels.
//
// Everything the kernels read lives in HBM as arrays of 16-byte words so that a
// lane fetches whole records with global_load_dwordx4:
//
//   wide node    128 B = 8 x float4   four exact child boxes in SoA form + four child words (what the builds make)
//   primitive     48 B = 3 x float4   one record per primitive, all types
//   compact node  64 B = 4 x float4   the four child boxes on a 256-step grid + four child words   } what the closest-hit
//   leaf block   128 B = 8 x float4   a leaf's exact guard box + its (up to two) primitive records } kernel reads
//   shade record  96 B = 6 x float4   a primitive's record + the vertex normals of a triangle (same index as its record)
//   material      32 B = 2 x float4
//
// Primitives are stored in the order in which the reference's recursive traversal
// (right child first, src/tracer/bvh.rs:148-155) reaches its leaves, so the array
// index of a primitive is its "visit rank"; equal-distance hits are resolved with
// it (see rt_traverse.hpp).
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

// ---- wide node -----------------------------------------------------------
// float4 0..5 : lo.x[4], lo.y[4], lo.z[4], hi.x[4], hi.y[4], hi.z[4]   (one lane of the float4 per child)
// float4 6    : child words (as raw bits)
// float4 7    : unused
//
// child word: RT_CHILD_NONE            empty slot
//             bit 31 clear             index of another wide node
//             bit 31 set               leaf: bits 28..30 = primitive count - 1, bits 0..27 = first primitive
#define RT_NODE_FLOAT4S 8
#define RT_CHILD_NONE 0xFFFFFFFFu
#define RT_CHILD_LEAF 0x80000000u
#define RT_LEAF_MAX_PRIMS 7u
#define RT_LEAF_FIRST_MASK 0x0FFFFFFFu

// ---- compact wide node and leaf block (the forms the closest-hit kernel reads) -----------------------------
// Compact node, 64 B = 4 x float4: the four child boxes on a 256-step grid of the node's own box.  Plane k of axis a
// lies at origin[a] + k * step[a], step a power of two; a child's box is rounded OUTWARD to the grid, so it contains
// the exact box.  A compact box only steers the traversal (any box that contains the exact one does, see
// rt_traverse.hpp); the exact box of a leaf slot is kept in the leaf's block as its guard and tested when the leaf is
// reached.
// float4 0 : origin.x, origin.y, origin.z, step.x
// float4 1 : step.y, step.z, lo.x[4 bytes, child c in byte c], lo.y[4]
// float4 2 : lo.z[4], hi.x[4], hi.y[4], hi.z[4]
// float4 3 : child words: an inner child's word is the index of its compact node; a leaf's word has bit 31 set,
//            bits 28..30 = primitive count - 1, bits 0..27 = index of its first leaf block
// Leaf block, 128 B = 8 x float4 (one line): up to two primitives of a leaf; a leaf with more takes consecutive blocks.
// float4 0 : guard lo.x, lo.y, lo.z, hi.x
// float4 1 : guard hi.y, hi.z, visit rank of the block's first primitive (bits), primitives in this block (bits)
// float4 2..4, 5..7 : the primitive records
#define RT_QNODE_FLOAT4S 4
#define RT_LEAF_BLOCK_FLOAT4S 8

// ---- primitive record ------------------------------------------------------
// word 10 = meta, word 11 = primitive id (scene declaration order)
// meta: bits 0..2 type, bit 3 flag (triangle: two_sided, boxes: hollow), bits 4..31 material index
//   triangle : w0..2 v0, w3..5 e1 = v1 - v0, w6..8 e2 = v2 - v0
//   sphere   : w0..2 center, w3 radius
//   abox     : w0..2 min, w3..5 max
//   cube     : w0..2 center, w3..5 half extents, w6..9 rotor (s, xy, xz, yz)
#define RT_PRIM_FLOAT4S 3
/* 16-byte words from one shade record to the next.  The record itself is 2 * RT_PRIM_FLOAT4S words (96 bytes); at
   8 words every record is one 128-byte line, so the stage that completes a hit asks for one line per path instead of
   1.5 on average (the builders write the records back to back and the scene spreads them out). */
#ifndef RT_SHADE_RECORD_FLOAT4S
#define RT_SHADE_RECORD_FLOAT4S 8
#endif
#define RT_PRIM_TRIANGLE 0u
#define RT_PRIM_SPHERE 1u
#define RT_PRIM_ABOX 2u
#define RT_PRIM_CUBE 3u
#define RT_PRIM_TYPE_MASK 7u
#define RT_PRIM_FLAG 8u
#define RT_PRIM_MAT_SHIFT 4

// LDS entries of the per-lane traversal stack; deeper entries go to an HBM overflow area.
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 11
#endif

struct RtDevMaterial {  // 32 B
    float albedo[3];
    int32_t type;
    float p0, p1;
    float pad[2];
};

struct RtDevCamera {
    float eye[3];
    float lens_radius;
    float horizontal[3];
    float pad0;
    float vertical[3];
    float pad1;
    float llc[3];
    float pad2;
    float u[3];
    float pad3;
    float v[3];
    float pad4;
};

// What a kernel needs to see of a scene (passed by value).
struct RtSceneView {
    const void *nodes;       // float4[n_nodes * 8]; a scene holds either these and prims, or qnodes and leaves
    const void *qnodes;      // float4[n_nodes * 4]: the compact nodes
    const void *leaves;      // float4[n_blocks * 8]: the leaf blocks
    const void *prims;       // float4[n_prims * 3]
    const void *shade_records; // float4[n_prims * RT_SHADE_RECORD_FLOAT4S]: the primitive's record and its three vertex normals
                               // side by side (96 bytes of a 128-byte line), one gather for the stage that completes a hit
    const void *materials;   // RtDevMaterial[n_materials]
    const float *env;        // rgb float triples or nullptr
    uint32_t env_w, env_h;
    uint32_t n_nodes, n_prims;
    float root_lo[3];
    float root_hi[3];
    uint32_t max_stack; // upper bound of traversal stack entries for this tree
    uint32_t pad;
};

#endif
