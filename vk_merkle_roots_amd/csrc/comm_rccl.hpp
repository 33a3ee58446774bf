// comm_rccl.hpp -- the multi-GPU exchange of the hot path: ONE all-gather of 32-byte sub-tree roots
// over RCCL (xGMI inside a node).  Included by vkmr_hip.hip (same translation unit as the rest of the
// C ABI: shares its error plumbing).
//
// The reference has no counterpart call: it drives one VkDevice and collects the slice roots on the
// host (src/vkmr/Reductions.cpp:125-145 reads each root back, :703-712 combines them in slice order
// through CpuSha256DforReductions, :56-69).  With slices sharded over GPUs that collection step is the
// gather below; the combine that follows is vkmr_hip_combine_async on one GPU.
//
// librccl (573 MB) is bound lazily, on the first communicator: a single-GPU run never loads it.  If
// the process already carries RCCL (PyTorch ships its own copy; a second copy in one process must be
// avoided) those symbols are used, otherwise librccl.so.1 is opened from the ROCm library path.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

namespace vkmr_comm_detail {

struct Rccl {
    bool ok = false;
    char why[256] = "";
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};

inline bool g_asked = false;   // rccl() has been called: only then does vkmr_hip_comm_info report on the binding

inline const Rccl& rccl()
{
    g_asked = true;
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;   // RTLD_DEFAULT first: RCCL already in the process (or a test double preloaded)
        auto sym = [&](const char* name) -> void* {
            void* p = h ? dlsym(h, name) : dlsym(RTLD_DEFAULT, name);
            return p;
        };
        if (!dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : names)
                if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
            if (!h) {
                snprintf(r.why, sizeof r.why, "librccl not loadable: %s", dlerror());
                return;
            }
        }
#define VKMR_BIND(field, name)                                                                     \
    r.field = reinterpret_cast<decltype(r.field)>(sym(name));                                      \
    if (!r.field) { snprintf(r.why, sizeof r.why, "librccl lacks %s", name); return; }
        VKMR_BIND(GetUniqueId, "ncclGetUniqueId")
        VKMR_BIND(CommInitRank, "ncclCommInitRank")
        VKMR_BIND(CommInitAll, "ncclCommInitAll")
        VKMR_BIND(CommDestroy, "ncclCommDestroy")
        VKMR_BIND(GroupStart, "ncclGroupStart")
        VKMR_BIND(GroupEnd, "ncclGroupEnd")
        VKMR_BIND(AllGather, "ncclAllGather")
        VKMR_BIND(GetErrorString, "ncclGetErrorString")
        VKMR_BIND(GetVersion, "ncclGetVersion")
#undef VKMR_BIND
        r.ok = true;
    });
    return r;
}

}  // namespace vkmr_comm_detail

// One communicator: the ranks this process drives (one per local device) out of `nranks` in all.
struct vkmr_comm_s {
    int nranks = 0;
    std::vector<int> devs;        // local devices, in local-rank order
    std::vector<int> ranks;       // their ranks in the communicator
    std::vector<ncclComm_t> comms;
};

static vkmr_status comm_fail(const char* what, ncclResult_t e)
{
    const auto& r = vkmr_comm_detail::rccl();
    snprintf(g_err, sizeof g_err, "%s: %s", what, (r.ok && r.GetErrorString) ? r.GetErrorString(e) : "RCCL error");
    return VKMR_ERR_COMM;
}

#define VKMR_NCCL(expr)                                         \
    do {                                                        \
        ncclResult_t e__ = (expr);                              \
        if (e__ != ncclSuccess) return comm_fail(#expr, e__);   \
    } while (0)

// out[k] = gathered[(k % nranks) * per_rank + k / nranks]: rank r holds, in order, the roots of slices
// r+1, r+1+nranks, ... (Slices::New places slice k on device (k-1) % D), the gather returns them rank by
// rank, the combine wants them in slice order (src/vkmr/Reductions.cpp:703-712).
__global__ __launch_bounds__(256) void roots_in_slice_order_kernel(const Node* __restrict__ gathered, uint32_t nranks, uint32_t per_rank,
                                                                   uint32_t total, Node* __restrict__ out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const Node n = vkmr_dev::load_node(gathered + (size_t)(k % nranks) * per_rank + k / nranks);
    vkmr_dev::store_node(out + k, n.w);
}

extern "C" {

vkmr_status vkmr_hip_comm_create_id(void* id)
{
    if (!id) return fail(VKMR_ERR_INVALID, "vkmr_hip_comm_create_id: null pointer");
    const auto& r = vkmr_comm_detail::rccl();
    if (!r.ok) return fail(VKMR_ERR_COMM, r.why);
    static_assert(sizeof(ncclUniqueId) == VKMR_COMM_ID_BYTES, "VKMR_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    ncclUniqueId uid;
    VKMR_NCCL(r.GetUniqueId(&uid));
    memcpy(id, &uid, sizeof uid);
    return VKMR_OK;
}

vkmr_status vkmr_hip_comm_init_rank(int dev, const void* id, int nranks, int rank, vkmr_comm* out)
{
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(VKMR_ERR_INVALID, "vkmr_hip_comm_init_rank: bad argument");
    const auto& r = vkmr_comm_detail::rccl();
    if (!r.ok) return fail(VKMR_ERR_COMM, r.why);
    VKMR_TRY(hipSetDevice(dev));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t c = nullptr;
    VKMR_NCCL(r.CommInitRank(&c, nranks, uid, rank));
    vkmr_comm_s* comm = new vkmr_comm_s;
    comm->nranks = nranks;
    comm->devs.push_back(dev);
    comm->ranks.push_back(rank);
    comm->comms.push_back(c);
    *out = comm;
    return VKMR_OK;
}

vkmr_status vkmr_hip_comm_init_all(const int* devs, int ndev, vkmr_comm* out)
{
    if (!devs || !out || ndev < 1) return fail(VKMR_ERR_INVALID, "vkmr_hip_comm_init_all: bad argument");
    const auto& r = vkmr_comm_detail::rccl();
    if (!r.ok) return fail(VKMR_ERR_COMM, r.why);
    vkmr_comm_s* comm = new vkmr_comm_s;
    comm->nranks = ndev;
    comm->comms.assign((size_t)ndev, nullptr);
    for (int i = 0; i < ndev; ++i) {
        comm->devs.push_back(devs[i]);
        comm->ranks.push_back(i);
    }
    const ncclResult_t e = r.CommInitAll(comm->comms.data(), ndev, devs);
    if (e != ncclSuccess) {
        delete comm;
        return comm_fail("ncclCommInitAll", e);
    }
    *out = comm;
    return VKMR_OK;
}

const char* vkmr_hip_comm_info(void)
{
    static thread_local char buf[512];
    if (!vkmr_comm_detail::g_asked) return "rccl=not loaded";   // asking would load librccl (573 MB) just to describe it
    const auto& r = vkmr_comm_detail::rccl();
    if (!r.ok) {
        snprintf(buf, sizeof buf, "rccl=unusable (%s)", r.why);
        return buf;
    }
    Dl_info info;
    const char* path = "?";
    if (dladdr(reinterpret_cast<void*>(r.AllGather), &info) && info.dli_fname) path = info.dli_fname;
    int ver = 0;
    if (r.GetVersion) (void)r.GetVersion(&ver);
    snprintf(buf, sizeof buf, "rccl=%s version=%d", path, ver);
    return buf;
}

vkmr_status vkmr_hip_comm_destroy(vkmr_comm c)
{
    if (!c) return VKMR_OK;
    const auto& r = vkmr_comm_detail::rccl();
    vkmr_status st = VKMR_OK;
    for (size_t i = 0; i < c->comms.size(); ++i) {
        if (!c->comms[i] || !r.ok) continue;
        (void)hipSetDevice(c->devs[i]);
        const ncclResult_t e = r.CommDestroy(c->comms[i]);
        if (e != ncclSuccess) st = comm_fail("ncclCommDestroy", e);
    }
    delete c;
    return st;
}

vkmr_status vkmr_hip_comm_size(vkmr_comm c, int* nranks, int* nlocal)
{
    if (!c) return fail(VKMR_ERR_INVALID, "vkmr_hip_comm_size: null communicator");
    if (nranks) *nranks = c->nranks;
    if (nlocal) *nlocal = (int)c->comms.size();
    return VKMR_OK;
}

vkmr_status vkmr_hip_gather_roots_async(vkmr_comm c, const vkmr_stream* streams, const vkmr_digest* const* roots_dev,
                                        uint32_t per_rank, vkmr_digest* const* all_dev)
{
    if (!c || !streams || !roots_dev || !all_dev || per_rank == 0) return fail(VKMR_ERR_INVALID, "vkmr_hip_gather_roots_async: bad argument");
    const auto& r = vkmr_comm_detail::rccl();
    if (!r.ok) return fail(VKMR_ERR_COMM, r.why);
    const size_t nlocal = c->comms.size();
    for (size_t i = 0; i < nlocal; ++i)
        if (!roots_dev[i] || !all_dev[i]) return fail(VKMR_ERR_INVALID, "vkmr_hip_gather_roots_async: null buffer");
    // one collective; a process that drives several ranks issues them as one group
    if (nlocal > 1) VKMR_NCCL(r.GroupStart());
    ncclResult_t first_err = ncclSuccess;
    for (size_t i = 0; i < nlocal; ++i) {
        (void)hipSetDevice(c->devs[i]);
        const ncclResult_t e = r.AllGather(roots_dev[i], all_dev[i], (size_t)per_rank * 8u, ncclUint32, c->comms[i], S(streams[i]));
        if (e != ncclSuccess && first_err == ncclSuccess) first_err = e;
    }
    if (nlocal > 1) {
        const ncclResult_t e = r.GroupEnd();
        if (e != ncclSuccess && first_err == ncclSuccess) first_err = e;
    }
    if (first_err != ncclSuccess) return comm_fail("ncclAllGather", first_err);
    return VKMR_OK;
}

vkmr_status vkmr_hip_roots_in_slice_order_async(int dev, vkmr_stream s, const vkmr_digest* gathered_dev, uint32_t nranks,
                                                uint32_t per_rank, uint32_t total, vkmr_digest* out_dev)
{
    if (!gathered_dev || !out_dev || nranks == 0 || per_rank == 0) return fail(VKMR_ERR_INVALID, "vkmr_hip_roots_in_slice_order_async: bad argument");
    if ((uint64_t)total > (uint64_t)nranks * per_rank) return fail(VKMR_ERR_INVALID, "vkmr_hip_roots_in_slice_order_async: more roots than were gathered");
    if (total == 0) return VKMR_OK;
    VKMR_TRY(hipSetDevice(dev));
    hipLaunchKernelGGL(roots_in_slice_order_kernel, dim3((total + 255u) / 256u), dim3(256), 0, S(s), reinterpret_cast<const Node*>(gathered_dev),
                       nranks, per_rank, total, reinterpret_cast<Node*>(out_dev));
    VKMR_TRY(hipGetLastError());
    return VKMR_OK;
}

}  // extern "C"
