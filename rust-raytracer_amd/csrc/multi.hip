// multi.hip -- several GPUs behind the C ABI: scene replicas, passes dealt round-robin, ONE sum of the
// device-resident accumulators at the end (the reference's passes are additive, src/render.rs:214, and the
// caller divides by the pass count, src/main.rs:288).
//
// Two shapes:
//   one process, n devices   rt_multi_gpu_init / rt_render_passes_multi / rt_reduce: one host thread per replica
//                            while rendering; the sum is an RCCL reduce over the communicators of
//                            ncclCommInitAll, or -- no RCCL, or replicas sharing a device -- peer copies to the
//                            first replica's device and an add kernel there
//   one process per device   rt_comm_*: ncclCommInitRank from an id the caller hands round, ncclReduce in place
//
// RCCL is looked up at run time (dlopen), never linked: a process that already holds an RCCL (PyTorch brings its
// own) must not get a second one, and a machine without RCCL still loads the library and uses the peer-copy sum.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtcore.h"

extern "C" int rt_set_error_message(int code, const char *msg);  // rtcore.hip

namespace {

int fail(int code, const std::string &msg) { return rt_set_error_message(code, msg.c_str()); }
#define RT_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ---- the few RCCL entry points used, bound at run time (types as in rccl.h) ----
struct NcclUniqueId {
    char internal[128];
};
typedef void *NcclComm;
enum { kNcclSuccess = 0, kNcclFloat32 = 7, kNcclSum = 0 };
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    int (*CommInitAll)(NcclComm *, int, const int *) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*CommAbort)(NcclComm) = nullptr;  // optional
    int (*Reduce)(const void *, void *, size_t, int, int, int, NcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;  // why it is not usable
    bool ok() const { return handle != nullptr; }
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl &rccl() {
    std::call_once(g_rccl_once, []() {
        if (const char *e = std::getenv("RTCORE_NO_RCCL")) {
            if (*e && *e != '0') {
                g_rccl.why = "RTCORE_NO_RCCL is set";
                return;
            }
        }
        // the soname first: a copy that is already in the process (PyTorch's) is found under it;
        // RTCORE_RCCL_LIB names the one file to try instead (a test points it at a file that does not exist)
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        const char *only = std::getenv("RTCORE_RCCL_LIB");
        void *h = nullptr;
        std::string last_error;
        for (const char *n : names) {
            if (only && *only) n = only;
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
            const char *e = dlerror();  // (reading the message clears it: one call per failure)
            last_error = e ? e : "";
            if (only && *only) break;
        }
        if (!h) {
            g_rccl.why = "librccl not found: " + last_error;
            return;
        }
        Rccl r;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
        r.Reduce = (decltype(r.Reduce))dlsym(h, "ncclReduce");
        r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.Reduce || !r.GroupStart || !r.GroupEnd) {
            g_rccl.why = "librccl lacks an expected symbol";
            return;
        }
        r.handle = h;
        g_rccl = r;
    });
    return g_rccl;
}
std::string nccl_msg(const char *what, int rc) {
    const Rccl &r = rccl();
    return std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
}

__global__ void __launch_bounds__(256) add_kernel(float *__restrict__ dst, const float *__restrict__ src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = dst[i] + src[i];
}

}  // namespace

struct RtMulti {
    std::vector<int> devices;
    std::vector<RtScene *> scenes;
    std::vector<hipStream_t> streams;
    std::vector<float *> accum;  // per replica, device-resident, grown on demand
    size_t accum_floats = 0;
    float *stage = nullptr;  // on devices[0]: landing buffer of the peer-copy sum
    size_t stage_floats = 0;
    std::vector<float> host;  // landing buffer of the one download per rt_render_passes_multi call (kept between calls)
    std::vector<NcclComm> comms;  // ncclCommInitAll over the devices, made on first use
    bool distinct = true;         // no device named twice
    int last_reduce = 0;          // 1 RCCL, 2 peer copies
};

struct RtComm {
    NcclComm comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

namespace {

int ensure_accum(RtMulti *m, size_t n_floats) {
    if (n_floats <= m->accum_floats) return RT_OK;
    for (size_t g = 0; g < m->devices.size(); g++) {
        RT_HIP(hipSetDevice(m->devices[g]));
        if (m->accum[g]) (void)hipFree(m->accum[g]);
        m->accum[g] = nullptr;
        if (hipMalloc((void **)&m->accum[g], n_floats * sizeof(float)) != hipSuccess) {
            m->accum_floats = 0;
            return fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc of an accumulator failed");
        }
    }
    m->accum_floats = n_floats;
    return RT_OK;
}

int reduce_peer(RtMulti *m, float *const *bufs, size_t n) {
    const int root = m->devices[0];
    RT_HIP(hipSetDevice(root));
    if (n > m->stage_floats) {
        if (m->stage) (void)hipFree(m->stage);
        m->stage = nullptr, m->stage_floats = 0;
        if (hipMalloc((void **)&m->stage, n * sizeof(float)) != hipSuccess)
            return fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc of the staging buffer failed");
        m->stage_floats = n;
    }
    // the other replicas' streams have been synchronised by the caller; everything below is ordered on the root's stream
    hipStream_t st = m->streams[0];
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 8);
    for (size_t g = 1; g < m->devices.size(); g++) {
        if (m->devices[g] == root)  // a replica on the root's own device (rehearsal on one GPU)
            RT_HIP(hipMemcpyAsync(m->stage, bufs[g], n * sizeof(float), hipMemcpyDeviceToDevice, st));
        else
            RT_HIP(hipMemcpyPeerAsync(m->stage, root, bufs[g], m->devices[g], n * sizeof(float), st));
        hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, st, bufs[0], (const float *)m->stage, n);
        RT_HIP(hipGetLastError());
    }
    RT_HIP(hipStreamSynchronize(st));
    m->last_reduce = 2;
    return RT_OK;
}

int reduce_rccl(RtMulti *m, float *const *bufs, size_t n) {
    const Rccl &r = rccl();
    if (!r.ok()) return fail(RT_ERR_DEVICE, "RCCL is not usable: " + r.why);
    if (!m->distinct) return fail(RT_ERR_INVALID_ARGUMENT, "an RCCL communicator needs distinct devices");
    const int nd = (int)m->devices.size();
    if (m->comms.empty()) {
        m->comms.assign((size_t)nd, nullptr);
        int rc = r.CommInitAll(m->comms.data(), nd, m->devices.data());
        if (rc != kNcclSuccess) {
            m->comms.clear();
            return fail(RT_ERR_DEVICE, nccl_msg("ncclCommInitAll", rc));
        }
    }
    // A failure inside the group leaves the communicators in an unknown state: the group is closed, the communicators
    // are given up (aborted where the library can, so nothing waits for the replicas that were never enqueued) and the
    // next call makes new ones or takes the peer-copy sum.
    auto give_up = [&](const char *what, int code) {
        for (NcclComm c : m->comms)
            if (c) (void)(r.CommAbort ? r.CommAbort(c) : r.CommDestroy(c));
        m->comms.clear();
        return fail(RT_ERR_DEVICE, nccl_msg(what, code));
    };
    for (int g = 0; g < nd; g++)
        if (!bufs[g]) return fail(RT_ERR_INVALID_ARGUMENT, "null accumulator");
    int rc = r.GroupStart();
    if (rc != kNcclSuccess) return give_up("ncclGroupStart", rc);
    int bad = kNcclSuccess;
    for (int g = 0; g < nd && bad == kNcclSuccess; g++) {
        if (hipSetDevice(m->devices[g]) != hipSuccess) {
            bad = -1;
            break;
        }
        bad = r.Reduce(bufs[g], bufs[g], n, kNcclFloat32, kNcclSum, 0, m->comms[(size_t)g], m->streams[(size_t)g]);
    }
    if (bad != kNcclSuccess) {
        // every replica's call must be enqueued before the group can complete: abort the communicators first, then
        // close the group (an aborted communicator's pending work returns instead of waiting)
        for (NcclComm c : m->comms)
            if (c && r.CommAbort) (void)r.CommAbort(c);
        (void)r.GroupEnd();
        if (!r.CommAbort)
            for (NcclComm c : m->comms)
                if (c) (void)r.CommDestroy(c);
        m->comms.clear();
        return fail(RT_ERR_DEVICE, bad == -1 ? std::string("hipSetDevice failed") : nccl_msg("ncclReduce", bad));
    }
    rc = r.GroupEnd();
    if (rc != kNcclSuccess) return give_up("ncclGroupEnd", rc);
    for (int g = 0; g < nd; g++) {
        RT_HIP(hipSetDevice(m->devices[g]));
        RT_HIP(hipStreamSynchronize(m->streams[(size_t)g]));
    }
    m->last_reduce = 1;
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_multi_gpu_init(const RtSceneDesc *desc, const int *devices, int n_devices, RtMulti **out) {
    if (!desc || !devices || !out || n_devices < 1 || n_devices > 64) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    *out = nullptr;
    int before = 0;
    RT_HIP(hipGetDevice(&before));
    RtMulti *m = new RtMulti();
    m->devices.assign(devices, devices + n_devices);
    m->scenes.assign((size_t)n_devices, nullptr);
    m->streams.assign((size_t)n_devices, nullptr);
    m->accum.assign((size_t)n_devices, nullptr);
    for (int i = 0; i < n_devices; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) m->distinct = false;
    int rc = rt_scene_create_replicas(desc, devices, n_devices, m->scenes.data());
    for (int g = 0; g < n_devices && rc == RT_OK; g++) {
        if (hipSetDevice(devices[g]) != hipSuccess || hipStreamCreateWithFlags(&m->streams[(size_t)g], hipStreamNonBlocking) != hipSuccess)
            rc = fail(RT_ERR_DEVICE, "stream creation failed");
        if (rc == RT_OK) rc = rt_init(devices[g]);
    }
    (void)hipSetDevice(before);
    if (rc != RT_OK) {
        const std::string msg = rt_last_error();
        rt_multi_gpu_destroy(m);
        return fail(rc, msg);
    }
    *out = m;
    return RT_OK;
}

int rt_multi_gpu_destroy(RtMulti *m) {
    if (!m) return RT_OK;
    int before = 0;
    (void)hipGetDevice(&before);
    if (!m->comms.empty()) {  // (communicators exist only if RCCL was loaded: no look-up otherwise)
        const Rccl &r = rccl();
        for (NcclComm c : m->comms)
            if (c && r.ok()) (void)r.CommDestroy(c);
    }
    for (size_t g = 0; g < m->devices.size(); g++) {
        (void)hipSetDevice(m->devices[g]);
        if (m->streams[g]) {
            (void)hipStreamSynchronize(m->streams[g]);
            (void)hipStreamDestroy(m->streams[g]);
        }
        if (m->accum[g]) (void)hipFree(m->accum[g]);
        if (m->scenes[g]) rt_scene_destroy(m->scenes[g]);
    }
    if (m->stage) {
        (void)hipSetDevice(m->devices[0]);
        (void)hipFree(m->stage);
    }
    (void)hipSetDevice(before);
    delete m;
    return RT_OK;
}

int rt_multi_gpu_count(const RtMulti *m) { return m ? (int)m->devices.size() : 0; }
RtScene *rt_multi_gpu_scene(RtMulti *m, int i) { return (m && i >= 0 && i < (int)m->scenes.size()) ? m->scenes[(size_t)i] : nullptr; }
int rt_multi_gpu_last_reduce(const RtMulti *m) { return m ? m->last_reduce : 0; }

int rt_reduce(RtMulti *m, float *const *d_bufs, size_t n_floats, int how) {
    if (!m || !d_bufs || how < 0 || how > 2) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    // one replica: nothing to add; an explicit request for RCCL still goes through it (a one-rank communicator's
    // in-place reduce), so that the path can be exercised on a single GPU
    if (n_floats == 0 || (m->devices.size() == 1 && how != 1)) return RT_OK;
    int before = 0;
    RT_HIP(hipGetDevice(&before));
    int rc;
    if (how == 1 || (how == 0 && m->distinct && rccl().ok()))
        rc = reduce_rccl(m, d_bufs, n_floats);
    else
        rc = reduce_peer(m, d_bufs, n_floats);
    (void)hipSetDevice(before);
    return rc;
}

int rt_render_passes_multi(RtMulti *m, const RtRenderParams *params, uint32_t first_pass, uint32_t n_passes, float *accum_rgb) {
    if (!m || !params || !accum_rgb) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (n_passes == 0) return RT_OK;
    const size_t n = (size_t)params->width * params->height * 3;
    int before = 0;
    RT_HIP(hipGetDevice(&before));
    int rc = ensure_accum(m, n);
    if (rc) {
        (void)hipSetDevice(before);
        return rc;
    }
    const size_t nd = m->devices.size();
    std::vector<int> rcs(nd, RT_OK);
    std::vector<std::string> msgs(nd);
    std::vector<std::thread> workers;
    // replica g renders passes first + g, first + g + nd, ...: the random streams are keyed by the global pass
    // index, so the frame does not depend on the number of replicas (up to the order of the final additions)
    for (size_t g = 0; g < nd; g++) {
        workers.emplace_back([&, g]() {
            int r = RT_OK;
            if (hipSetDevice(m->devices[g]) != hipSuccess) r = fail(RT_ERR_DEVICE, "hipSetDevice failed");
            if (r == RT_OK && hipMemsetAsync(m->accum[g], 0, n * sizeof(float), m->streams[g]) != hipSuccess)
                r = fail(RT_ERR_DEVICE, "hipMemsetAsync failed");
            for (uint32_t k = (uint32_t)g; k < n_passes && r == RT_OK; k += (uint32_t)nd) {
                RtRenderParams p = *params;
                p.pass_index = first_pass + k;
                r = rt_render_pass_device(m->scenes[g], &p, m->accum[g], (void *)m->streams[g]);
            }
            if (r == RT_OK && hipStreamSynchronize(m->streams[g]) != hipSuccess) r = fail(RT_ERR_DEVICE, "hipStreamSynchronize failed");
            rcs[g] = r;
            if (r != RT_OK) msgs[g] = rt_last_error();  // the message is per thread: hand it to the caller's
        });
    }
    for (auto &w : workers) w.join();
    for (size_t g = 0; g < nd; g++)
        if (rcs[g] != RT_OK) {
            (void)hipSetDevice(before);
            return fail(rcs[g], msgs[g]);
        }
    rc = rt_reduce(m, m->accum.data(), n, 0);
    if (rc == RT_OK) {
        // the one PCIe transfer of the render: the summed frame comes down and is added to the caller's buffer
        if (m->host.size() < n) m->host.resize(n);
        const float *host = m->host.data();
        if (hipSetDevice(m->devices[0]) != hipSuccess ||
            hipMemcpy(m->host.data(), m->accum[0], n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(RT_ERR_DEVICE, "download of the summed frame failed");
        else
            for (size_t i = 0; i < n; i++) accum_rgb[i] = accum_rgb[i] + host[i];
    }
    (void)hipSetDevice(before);
    return rc;
}

int rt_multi_gpu_stats(RtMulti *m, RtStats *total) {
    if (!m || !total) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    std::memset(total, 0, sizeof(*total));
    int before = 0;
    RT_HIP(hipGetDevice(&before));
    for (size_t g = 0; g < m->devices.size(); g++) {
        RtStats st;
        int rc = rt_scene_stats(m->scenes[g], &st);
        if (rc) {
            (void)hipSetDevice(before);
            return rc;
        }
        total->rays += st.rays, total->paths += st.paths, total->node_visits += st.node_visits, total->prim_tests += st.prim_tests;
        total->extend_ms += st.extend_ms, total->shade_ms += st.shade_ms, total->other_ms += st.other_ms;
        total->extend_launches += st.extend_launches;
        total->node_steps += st.node_steps, total->node_step_lanes += st.node_step_lanes;
        total->leaf_steps += st.leaf_steps, total->leaf_step_lanes += st.leaf_step_lanes, total->l1_lookups += st.l1_lookups;
    }
    (void)hipSetDevice(before);
    return RT_OK;
}

// ---- one process per device ----
int rt_comm_available(void) { return rccl().ok() ? 1 : 0; }

int rt_comm_unique_id(unsigned char id[128]) {
    if (!id) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    const Rccl &r = rccl();
    if (!r.ok()) return fail(RT_ERR_DEVICE, "RCCL is not usable: " + r.why);
    NcclUniqueId u;
    int rc = r.GetUniqueId(&u);
    if (rc != kNcclSuccess) return fail(RT_ERR_DEVICE, nccl_msg("ncclGetUniqueId", rc));
    std::memcpy(id, u.internal, 128);
    return RT_OK;
}

int rt_comm_init_rank(const unsigned char id[128], int rank, int world, RtComm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    *out = nullptr;
    const Rccl &r = rccl();
    if (!r.ok()) return fail(RT_ERR_DEVICE, "RCCL is not usable: " + r.why);
    NcclUniqueId u;
    std::memcpy(u.internal, id, 128);
    int device = 0;
    RT_HIP(hipGetDevice(&device));
    RtComm *c = new RtComm();
    c->rank = rank, c->world = world, c->device = device;
    int rc = r.CommInitRank(&c->comm, world, u, rank);
    if (rc != kNcclSuccess) {
        delete c;
        return fail(RT_ERR_DEVICE, nccl_msg("ncclCommInitRank", rc));
    }
    *out = c;
    return RT_OK;
}

int rt_comm_reduce(RtComm *c, float *d_buf, size_t n_floats, int root, void *stream) {
    if (!c || !d_buf || root < 0 || root >= c->world) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    const Rccl &r = rccl();
    if (!r.ok() || !c->comm) return fail(RT_ERR_DEVICE, "RCCL is not usable: " + r.why);
    int rc = r.Reduce(d_buf, d_buf, n_floats, kNcclFloat32, kNcclSum, root, c->comm, (hipStream_t)stream);
    if (rc != kNcclSuccess) return fail(RT_ERR_DEVICE, nccl_msg("ncclReduce", rc));
    return RT_OK;
}

int rt_comm_destroy(RtComm *c) {
    if (!c) return RT_OK;
    const Rccl &r = rccl();
    if (c->comm && r.ok()) (void)r.CommDestroy(c->comm);
    delete c;
    return RT_OK;
}

}  // extern "C"
