// gpu_build.hpp -- the opt-in scene build on the device (gpu_build.hip): primitive records, Morton-ordered
// binary radix tree and 4-wide collapse without a host sort.  Builds on the CURRENT device.
#ifndef RT_GPU_BUILD_HPP
#define RT_GPU_BUILD_HPP

#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/rtcore.h"

namespace rtcore {

struct DeviceBuild {
    // device allocations (hipMalloc) that the caller takes over; each has 128 spare bytes at its end
    void *nodes = nullptr, *prims = nullptr, *shade_records = nullptr;
    size_t nodes_bytes = 0, prims_bytes = 0, shade_bytes = 0;
    uint64_t n_prims = 0, n_triangles = 0, n_nodes = 0, n_binary_nodes = 0;
    uint32_t wide_depth = 0, max_stack = 0;
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
    double ms_upload = 0.0, ms_total = 0.0;  // host clock: description packed and uploaded / whole build
    void release();                          // frees the three allocations (for a caller that does not take them)
};

// Returns RT_OK or a negative RtStatus and fills `err`.  On failure nothing stays allocated.
int build_device_scene(const RtSceneDesc *desc, DeviceBuild &out, std::string &err);

}  // namespace rtcore
#endif
