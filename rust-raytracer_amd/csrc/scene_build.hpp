// scene_build.hpp -- host side of rt_scene_create: builds the reference-shaped binary
// tree, numbers the primitives in the reference's visit order and collapses the tree
// into 4-wide nodes.  No device code here.
#ifndef RT_SCENE_BUILD_HPP
#define RT_SCENE_BUILD_HPP

#include <cstdint>
#include <memory>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rtcore.h"
#include "rt_types.h"

namespace rtcore {

// std::vector::resize without the zero fill, for large arrays of which every element is written right after
// (the fill of 2.6 GB was a quarter of the build of a 10 M-triangle scene).
template <class T>
struct NoInitAllocator : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = NoInitAllocator<U>;
    };
    template <class U>
    void construct(U *p) noexcept {
        ::new (static_cast<void *>(p)) U;
    }
    template <class U, class... Args>
    void construct(U *p, Args &&...args) {
        ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...);
    }
};
template <class T>
using RawVector = std::vector<T, NoInitAllocator<T>>;

struct HostScene {
    RawVector<float> nodes;          // 32 floats per wide node
    RawVector<uint32_t> prims;       // 12 words per primitive, visit-rank order
    RawVector<float> shade_records;  // 24 floats per primitive: its record again, then three vertex normals (zero for non-triangles)
    std::vector<RtDevMaterial> materials;
    std::vector<float> env;
    uint32_t env_w = 0, env_h = 0;
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
    uint64_t n_prims = 0, n_triangles = 0, n_nodes = 0, n_binary_nodes = 0;
    uint32_t wide_depth = 0, max_stack = 0;
    RtCamera camera{};
};

// Returns RT_OK or a negative RtStatus and fills `err`.
int build_host_scene(const RtSceneDesc *desc, HostScene &out, std::string &err);

// Pieces of the build that the device-side builder shares with it (same arithmetic, hence the same records).
// (check_indices = false leaves the vertex indices of the meshes unchecked: the device build checks them itself)
int check_scene_desc(const RtSceneDesc *desc, uint64_t &n_primitives, std::string &err, bool check_indices = true);
int fill_materials_camera_env(const RtSceneDesc *desc, HostScene &out, std::string &err);
void simple_object_record(const RtObject &o, uint32_t id, uint32_t *rec12, float *nrm12, float *box6);
void mesh_rotor(const RtMesh &m, float rot[4]);

// Camera::new (src/camera.rs:21-50)
void make_camera(const RtCamera &c, float aspect, RtDevCamera &out);

// The tree shapes that were measured against the default and lost -- a rebuilt (surface-area) interior, the greedy
// 4-wide collapse, the Morton-order tree of the device build -- answer to their environment switches only in a library
// built with -DRTCORE_EXPERIMENTS (`make experiments`: librtcore_experiments.so, which the tests that check them load
// through RTCORE_LIB).  In the shipped library the switches read as unset, so one tree shape per builder is all that
// every kernel change has to keep exact.
inline const char *experiment_env(const char *name) {
#ifdef RTCORE_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

}  // namespace rtcore
#endif
