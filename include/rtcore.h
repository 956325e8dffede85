/*
 * rtcore.h -- C ABI of the MI355X path-tracing core (librtcore.so).
 *
 * The core replaces the CPU hot path of Teetlez/Rust-Raytracer behind plain C
 * entry points: closest-hit BVH traversal over spheres, axis boxes, rotated
 * boxes, triangles and meshes, the bounce loop with the four scatter models,
 * and the environment lookup.  Everything is float32; buffers are row-major,
 * top image row first, exactly as the reference's `Vec<Vec3>` frame buffers.
 *
 * The reference has no FFI of its own.  Each entry point names the Rust item a
 * host-side `extern "C"` block would call it in place of (file:line relative to
 * the reference repository root).
 *
 * Conventions
 *   - every function returns RT_OK (0) or a negative RtStatus; nothing throws
 *     or aborts across this boundary; rt_last_error() gives the message of the
 *     last failure on the calling thread;
 *   - the caller owns every buffer it passes; the core copies what it keeps;
 *   - a scene is immutable after rt_scene_create (the reference shares its
 *     world as Arc<Bvh>, src/render.rs:155);
 *   - one host thread drives one scene at a time; a scene lives on the HIP
 *     device that was current when it was created;
 *   - "_device" variants take device pointers and a hipStream_t (as void*) and
 *     only enqueue work: no host synchronisation, except that the first call on
 *     a scene (and any later call that needs larger work buffers than every call
 *     before it) allocates those buffers and asks the device once for its free
 *     memory; the others take host pointers, copy and synchronise.
 *
 * prim_id: the reference's HitRecord carries no primitive identity
 * (src/tracer/hittable.rs:13-19).  Here a primitive id is the index in scene
 * declaration order: object i of RtSceneDesc::objects contributes one id, except
 * a mesh object, which contributes one id per face, in face order.  Ids are
 * assigned in object order with no gaps; -1 means "no hit".
 *
 * Equal distances: when two primitives are hit at exactly the same t, the
 * reference's winner follows from the order in which Bvh::hit reaches them
 * (right child first, src/tracer/bvh.rs:148-157) and from each primitive's own
 * bound check (a triangle and a box entry face accept t == t_max, a sphere and a
 * box exit face do not).  The core returns the winner of that rule for the tree
 * ITS build produces.  Bvh::new sorts with a comparator that is not a strict weak
 * order (src/tracer/bvh.rs:120-139), so the tree, and with it the winner among
 * exactly coincident hits, depends on the sort algorithm: the core uses a plain
 * stable top-down merge sort, Rust's slice::sort_by is a different stable merge
 * sort (run-detecting; driftsort from 1.81), and the two are not proven to give
 * the same permutation.  t is the same either way; only which of several
 * primitives at one identical t is named can differ.
 */
#ifndef RTCORE_H
#define RTCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum RtStatus {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_OUT_OF_MEMORY = -2,
    RT_ERR_DEVICE = -3,      /* a HIP call failed or no device is usable */
    RT_ERR_EMPTY_SCENE = -4, /* the reference panics on an empty world (src/tracer/bvh.rs:45) */
    RT_ERR_INTERNAL = -5
} RtStatus;

/* ---- materials: src/material.rs:138-144, constructors :147-174 ---- */
typedef enum RtMaterialType {
    RT_MAT_DIELECTRIC = 0, /* albedo = absorption, p0 = refractive_index, p1 = roughness */
    RT_MAT_LAMBERTIAN = 1, /* albedo */
    RT_MAT_METAL = 2,      /* albedo, p0 = roughness */
    RT_MAT_GLOSSY = 3      /* albedo, p0 = reflectance, p1 = roughness */
} RtMaterialType;

typedef struct RtMaterial {
    int32_t type; /* RtMaterialType */
    float albedo[3];
    float p0;
    float p1;
} RtMaterial;

/* ---- objects: the reference's Hittable implementors (src/tracer/) ---- */
typedef enum RtObjectType {
    RT_OBJ_SPHERE = 0,   /* Sphere::new(center, radius, mat)        src/tracer/sphere.rs:18 */
    RT_OBJ_ABOX = 1,     /* ABox::new(center, size, mat)            src/tracer/cube.rs:20   */
    RT_OBJ_CUBE = 2,     /* Cube::new(center, size, rotation, mat)  src/tracer/cube.rs:90   */
    RT_OBJ_TRIANGLE = 3, /* Triangle::new(vertices, normals, two_sided, mat) src/tracer/triangle.rs:21 */
    RT_OBJ_MESH = 4      /* Mesh::new(polygons, translation, scale, rotation, cull, mat) src/tracer/mesh.rs:22 */
} RtObjectType;

/*
 * One scene object.  Field use by type:
 *   SPHERE   a = center, b[0] = radius
 *   ABOX     a = center, b = size (any negative component makes the box hollow)
 *   CUBE     a = center, b = size, c = rotation in units of pi, as the scene file
 *            gives it (Cube::new multiplies by pi itself, src/tracer/cube.rs:99)
 *   TRIANGLE tri_vertices / tri_normals (3x3 each), flag = two_sided
 *   MESH     mesh = index into RtSceneDesc::meshes
 */
typedef struct RtObject {
    int32_t type;     /* RtObjectType */
    int32_t material; /* index into RtSceneDesc::materials */
    float a[3];
    float b[3];
    float c[3];
    float tri_vertices[9];
    float tri_normals[9];
    int32_t flag;
    int32_t mesh;
} RtObject;

/*
 * One triangle mesh = one tobj model as Mesh::new receives it
 * (src/io.rs:175-184, src/tracer/mesh.rs:22-87): unified indexing, three
 * indices per face, a normal for every position.  `rotation` is in radians,
 * i.e. already multiplied by pi as src/io.rs:180 does.  The core applies
 * v' = rot(v * scale) + translation and n' = rot(n) itself.
 */
typedef struct RtMesh {
    const float *positions; /* 3 * n_vertices */
    const float *normals;   /* 3 * n_vertices */
    const uint32_t *indices; /* 3 * n_faces */
    uint64_t n_vertices;
    uint64_t n_faces;
    float translation[3];
    float scale[3];
    float rotation[3];
    int32_t cull_backface; /* Triangle::two_sided = !cull_backface (src/tracer/mesh.rs:78) */
    int32_t reserved;
} RtMesh;

/* Camera::new arguments (src/camera.rs:21-29); vup is fixed to +y as in src/io.rs:198. */
typedef struct RtCamera {
    float eye[3];
    float lookat[3];
    float vup[3];
    float fov_degrees;
    float aperture;
    float focus_dist;
} RtCamera;

/* radiant::Image as the renderer reads it (src/render.rs:87-104): row 0 is "up". */
typedef struct RtEnvMap {
    const float *rgb; /* width * height * 3, NULL = procedural gradient sky */
    uint32_t width;
    uint32_t height;
} RtEnvMap;

typedef struct RtSceneDesc {
    const RtMaterial *materials;
    uint64_t n_materials;
    const RtObject *objects;
    uint64_t n_objects;
    const RtMesh *meshes;
    uint64_t n_meshes;
    RtCamera camera;
    RtEnvMap env;
} RtSceneDesc;

typedef struct RtScene RtScene; /* opaque */

/* Renderer::render / Renderer::preview modes (src/render.rs:31-36). */
typedef enum RtMode { RT_MODE_NORMALS = 0, RT_MODE_COLORS = 1, RT_MODE_IMAGE = 2 } RtMode;

/*
 * The Renderer fields that are not part of the world (src/render.rs:151-160)
 * plus what replaces the reference's unseeded generators: every random number
 * is a pure function of (seed, pixel, pass_index * samples + sample, bounce,
 * dimension), so a pass gives the same image on any device count.
 */
typedef struct RtRenderParams {
    uint32_t width;
    uint32_t height;
    uint32_t samples;     /* Renderer::sample_rate */
    uint32_t max_bounce;  /* Renderer::max_bounce */
    float light_clamp;    /* Renderer::light_clamp; INFINITY = no clamp */
    int32_t mode;         /* RtMode */
    uint32_t pass_index;  /* global index of this pass (0-based) */
    uint32_t seed;
    uint64_t batch_paths; /* paths in flight per wavefront batch; 0 = default */
    /* Render only pixels [first_pixel, first_pixel + pixel_count) of the frame (row-major from the top row, the
     * index p of src/render.rs:163-176); pixel_count 0 = the whole frame.  The buffer is the whole frame either
     * way.  Pixels are independent within a pass, so ranges split one pass over several GPUs. */
    uint64_t first_pixel;
    uint64_t pixel_count;
} RtRenderParams;

typedef struct RtSceneInfo {
    uint64_t n_primitives;
    uint64_t n_triangles;
    uint64_t n_wide_nodes;
    uint64_t n_binary_nodes; /* nodes of the reference-shaped tree the wide tree was collapsed from */
    uint32_t wide_depth;
    uint32_t node_bytes;      /* bytes of one wide node as the closest-hit kernel reads it (64: compact form; 128 with RTCORE_NODES=wide) */
    uint32_t primitive_bytes; /* bytes of one primitive record */
    uint32_t builder;         /* RtBuilder that made the tree (RT_BUILD_EXACT or RT_BUILD_DEVICE) */
    uint64_t device_bytes; /* HBM held by the scene */
    float bounds_min[3];
    float bounds_max[3];
    double build_ms; /* host clock of the build: description in, tree on the device (0 from rt_scene_validate) */
} RtSceneInfo;

/* Counters of the last render pass / closest-hit batch on a scene. */
typedef struct RtStats {
    uint64_t rays;        /* Bvh::hit invocations: primary rays + every continued bounce */
    uint64_t paths;       /* camera paths started */
    uint64_t node_visits; /* only filled by the *_counted entry points, else 0 */
    uint64_t prim_tests;
    double extend_ms; /* device time in the closest-hit stage, when timing was requested */
    double shade_ms;
    double other_ms;
    uint64_t extend_launches; /* closest-hit kernel launches the times above cover */
    /* Lane occupancy of the closest-hit kernel's steps, only filled by the *_counted entry points / rt_set_counting:
     * a wave runs one step at a time, for the lanes whose ray is at a node (node step) or at a leaf (leaf step);
     * *_lanes / (64 * *_steps) is the fraction of a wave's lanes a step of that kind works for. */
    uint64_t node_steps;
    uint64_t node_step_lanes;
    uint64_t leaf_steps;
    uint64_t leaf_step_lanes;
    /* Line look-ups the node and primitive fetches of the closest-hit kernel ask of the vector L1 (one per load
     * instruction and different 128-byte line among its lanes); only filled by the counting kernel. */
    uint64_t l1_lookups;
} RtStats;

/* Library / device bring-up.  device < 0 keeps the current HIP device. */
int rt_init(int device);
const char *rt_last_error(void);
const char *rt_version(void);

/*
 * Builds the world.  Replaces io::load_scene's object loop + Bvh::new
 * (src/io.rs:112-190, src/tracer/bvh.rs:39-81) and Camera::new
 * (src/camera.rs:21-50; aspect ratio comes from RtRenderParams at render time,
 * as src/io.rs:200 takes it from the arguments).
 */
int rt_scene_create(const RtSceneDesc *desc, RtScene **out);
/* The same build once, uploaded to every device of `devices` (a device may be named more than once); out[n]. */
int rt_scene_create_replicas(const RtSceneDesc *desc, const int *devices, int n, RtScene **out);
/*
 * Which build makes the tree.
 *   RT_BUILD_EXACT   (the default) restates Bvh::new on the host (src/tracer/bvh.rs:39-117): the reference's tree,
 *                    hence the reference's winner among hits at exactly the same t (see "Equal distances" above).
 *   RT_BUILD_DEVICE  builds on the GPU: primitive boxes, the reference's median splits level by level (no sort), the same
 *                    4-wide collapse.  Primitive records are the same bits as the exact build's, so (prim_id, t,
 *                    normal) are the same for every ray whose closest hit is not an exact tie between two primitives
 *                    (which of the two is named then follows this tree's order, not the reference's) and that does not
 *                    graze a leaf's guard box within rounding.  Tens of times faster on large meshes, same traversal
 *                    speed.
 *   RT_BUILD_DEFAULT RT_BUILD_DEVICE when the environment has RTCORE_BUILD=gpu, else RT_BUILD_EXACT.
 * rt_scene_create(desc, out) is rt_scene_create_with(desc, RT_BUILD_DEFAULT, out).
 */
typedef enum RtBuilder { RT_BUILD_DEFAULT = 0, RT_BUILD_EXACT = 1, RT_BUILD_DEVICE = 2 } RtBuilder;
int rt_scene_create_with(const RtSceneDesc *desc, int builder, RtScene **out);
int rt_scene_create_replicas_with(const RtSceneDesc *desc, int builder, const int *devices, int n, RtScene **out);
/* Host-only dry run of rt_scene_create: validates the description, builds and collapses the
 * tree, fills `info` (device_bytes = what the scene would hold) and frees everything.  Needs no device. */
int rt_scene_validate(const RtSceneDesc *desc, RtSceneInfo *info);
int rt_scene_destroy(RtScene *scene);
int rt_scene_info(const RtScene *scene, RtSceneInfo *info);
int rt_scene_stats(const RtScene *scene, RtStats *stats);

/*
 * One pass: accum[p] += (sum over `samples` of radiance) / samples for every
 * pixel p.  Replaces Renderer::render(&self, buffer, mode) (src/render.rs:162),
 * called once per pass from src/main.rs:262.  `accum_rgb` is width*height*3
 * floats, in/out.
 */
int rt_render_pass(RtScene *scene, const RtRenderParams *params, float *accum_rgb);
int rt_render_pass_device(RtScene *scene, const RtRenderParams *params, float *d_accum_rgb,
                          void *stream);

/* Renderer::preview(&self, mode) (src/render.rs:222): 1 sample, no bounce; out is overwritten. */
int rt_preview(RtScene *scene, const RtRenderParams *params, float *out_rgb);
int rt_preview_device(RtScene *scene, const RtRenderParams *params, float *d_out_rgb, void *stream);

/*
 * <Bvh as Hittable>::hit(&self, &Ray, t_min, t_max) (src/tracer/bvh.rs:143) on a
 * batch.  org_dir holds n rays as 6 floats (origin, direction); directions are
 * normalised on entry as Ray::new does (src/ray.rs:10-15).  Outputs (any may be
 * NULL): prim_id[n] (-1 = miss), t[n], normal[3n], point[3n]; t/normal/point of
 * a miss are 0.
 */
int rt_closest_hit_batch(RtScene *scene, const float *org_dir, float t_min, float t_max, size_t n,
                         int32_t *prim_id, float *t, float *normal, float *point);
int rt_closest_hit_batch_device(RtScene *scene, const float *d_org_dir, float t_min, float t_max,
                                size_t n, int32_t *d_prim_id, float *d_t, float *d_normal,
                                float *d_point, void *stream);
/* Same result; additionally counts node visits and primitive tests into RtStats (slower). */
int rt_closest_hit_batch_counted_device(RtScene *scene, const float *d_org_dir, float t_min,
                                        float t_max, size_t n, int32_t *d_prim_id, float *d_t,
                                        float *d_normal, float *d_point, void *stream);

/*
 * rt_closest_hit_batch_device for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same kernels and
 * same results as rt_closest_hit_batch_device; the work memory (packed rays, hit records, the launch's work
 * counter, the overflow area of the traversal stacks) is a 64 MiB array that each GPU the library runs on
 * holds from load time until the process ends, so the call issues no hipMalloc / hipFree and creates no events.
 * Cost: those 64 MiB of HBM are taken on EVERY GPU of every process that loads the library, whether or not this
 * entry point is ever called (the array is part of the library's device code).
 * Limits: n <= RT_NOALLOC_MAX_RAYS per call (RT_ERR_INVALID_ARGUMENT beyond it); the launch uses fewer blocks
 * when the scene's tree is deep, so that the stacks' overflow area fits.  RtStats is not updated.  Every call
 * on a GPU uses that GPU's one array: calls that overlap on one GPU (two streams, two threads) race; calls on
 * one stream are ordered and safe.
 */
#define RT_NOALLOC_MAX_RAYS 262144u
int rt_closest_hit_batch_noalloc_device(RtScene *scene, const float *d_org_dir, float t_min, float t_max,
                                        size_t n, int32_t *d_prim_id, float *d_t, float *d_normal,
                                        float *d_point, void *stream);

/*
 * Several GPUs.  The reference renders its passes one after another on one host and adds them up
 * (src/main.rs:256-288, src/render.rs:214); passes are independent, so they are dealt round-robin to replicas of the
 * scene, one per GPU, each adding into an accumulator that stays on its GPU, and the accumulators are summed ONCE
 * when the passes are done.
 *
 * One process, n GPUs (the shape of the reference's host):
 *   rt_multi_gpu_init        one host build, one replica per entry of `devices` (an entry may repeat: several
 *                            replicas on one GPU rehearse the path on a one-GPU machine)
 *   rt_render_passes_multi   passes first_pass .. first_pass + n_passes - 1 (params->pass_index is ignored), replica g
 *                            takes first_pass + g, + g + n, ...; one host thread per replica; then rt_reduce, one
 *                            download, and accum_rgb[i] += sum[i].  The frame equals the one-GPU frame of the same
 *                            passes up to the order of the float additions across replicas.
 *   rt_reduce                d_bufs[0] += d_bufs[1] + ... (d_bufs[g] on devices[g], n_floats each; work on them must
 *                            have completed).  how: 0 = RCCL reduce over xGMI when the RCCL library can be loaded and
 *                            the devices are distinct, else peer copies + an add kernel on devices[0]; 1 = RCCL or
 *                            fail; 2 = peer copies.  Synchronous.  rt_multi_gpu_last_reduce tells which ran (1 / 2).
 * One process per GPU (launched by torchrun, MPI, ...):
 *   rt_comm_unique_id on one rank, the 128 bytes handed to every rank by the caller's own means,
 *   rt_comm_init_rank on every rank with its GPU current, rt_comm_reduce (in place, sum, result on `root`,
 *   enqueued on `stream`, no host synchronisation), rt_comm_destroy.
 * RCCL is loaded at run time (librccl.so.1); without it everything but how = 1 and rt_comm_* still works.
 */
typedef struct RtMulti RtMulti;
typedef struct RtComm RtComm;
int rt_multi_gpu_init(const RtSceneDesc *desc, const int *devices, int n_devices, RtMulti **out);
int rt_multi_gpu_destroy(RtMulti *multi);
int rt_multi_gpu_count(const RtMulti *multi);
RtScene *rt_multi_gpu_scene(RtMulti *multi, int i);
int rt_render_passes_multi(RtMulti *multi, const RtRenderParams *params, uint32_t first_pass, uint32_t n_passes,
                           float *accum_rgb);
int rt_reduce(RtMulti *multi, float *const *d_bufs, size_t n_floats, int how);
int rt_multi_gpu_last_reduce(const RtMulti *multi);
int rt_multi_gpu_stats(RtMulti *multi, RtStats *total); /* sum over the replicas; resets their counters */
int rt_comm_available(void);
int rt_comm_unique_id(unsigned char id[128]);
int rt_comm_init_rank(const unsigned char id[128], int rank, int world, RtComm **out);
int rt_comm_reduce(RtComm *comm, float *d_buf, size_t n_floats, int root, void *stream);
int rt_comm_destroy(RtComm *comm);

/* When nonzero, render passes time their stages with HIP events into RtStats (adds syncs). */
int rt_set_stage_timing(RtScene *scene, int enabled);

/* When nonzero, render passes use the counting closest-hit kernel: RtStats.node_visits / prim_tests
 * are filled (slower; for measuring the bytes a ray touches, not for production). */
int rt_set_counting(RtScene *scene, int enabled);

/*
 * Display side, on the device (no scene needed; device pointers, no host synchronisation).
 * rt_tonemap_device: argb[i] = to_rgb(&(rgb[i] / divisor), inv_gamma), the per-pass display transform of
 * src/main.rs:266-271 (divisor = passes so far) and the final one of src/main.rs:134-137 (divisor = 1).
 * rt_bilateral_filter_device: one iteration of bilateral_filter over the frame (src/filter.rs:14-48); the
 * reference runs three, with (diameter, sigma_i, sigma_s) = (9/i, 0.05/i, 1/i), i = 1..3 (src/main.rs:114-132).
 * in and out must not alias.
 */
int rt_tonemap_device(const float *d_rgb, size_t n_pixels, float divisor, float inv_gamma, uint32_t *d_argb,
                      void *stream);
int rt_bilateral_filter_device(const float *d_in, float *d_out, uint32_t width, uint32_t height, uint32_t diameter,
                               float sigma_i, float sigma_s, void *stream);
/* The same on host buffers (copied over PCIe, synchronous), for a host that keeps its frame in a Vec<Vec3>
 * as src/main.rs does. */
int rt_tonemap(const float *rgb, size_t n_pixels, float divisor, float inv_gamma, uint32_t *argb);
int rt_bilateral_filter(const float *in, float *out, uint32_t width, uint32_t height, uint32_t diameter,
                        float sigma_i, float sigma_s);

/*
 * Host-side helpers that need no device: the reference's display transform,
 * kept here so a host without the Rust crate can drain the core.
 * to_rgb(&Vec3, gamma) -> 0xFFRRGGBB (src/render.rs:39-55).
 */
uint32_t rt_to_rgb(const float rgb[3], float inv_gamma);

#ifdef __cplusplus
}
#endif
#endif /* RTCORE_H */
