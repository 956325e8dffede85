// compact.hpp -- the form of a scene's tree that the closest-hit kernel reads (compact.hip): 64-byte nodes whose child
// boxes lie on a 256-step grid of the node's own box, and 128-byte leaf blocks that hold a leaf's exact guard box next
// to its primitive records (rt_types.h has both layouts).  Made on the CURRENT device from the 128-byte nodes and the
// primitive records that either build leaves there.
#ifndef RT_COMPACT_HPP
#define RT_COMPACT_HPP

#include <cstddef>
#include <cstdint>
#include <string>

namespace rtcore {

struct CompactTree {
    // device allocations (hipMalloc) that the caller takes over; each has 128 spare bytes at its end
    void *qnodes = nullptr, *leaves = nullptr;
    size_t qnodes_bytes = 0, leaves_bytes = 0;
    uint64_t n_blocks = 0;
    void release();
};

// Returns RT_OK or a negative RtStatus and fills `err`; on failure nothing stays allocated.  RT_ERR_INVALID_ARGUMENT
// with an empty `out` when the leaf blocks would outgrow what a child word can address (the caller keeps the 128-byte
// nodes then).
int compact_tree(const void *nodes, uint64_t n_nodes, const void *prims, uint64_t n_prims, CompactTree &out, std::string &err);

}  // namespace rtcore
#endif
