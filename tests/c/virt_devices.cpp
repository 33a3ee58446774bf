// virt_devices.cpp -- TEST DOUBLE, loaded with LD_PRELOAD by tests/test_frontend.py: makes the one GPU
// of the test box look like a node of VKMR_TEST_VIRTUAL_DEVICES GPUs to the UNMODIFIED product binaries
// (libvkmr_hip.so, vkmr), so that the multi-device host paths of "hip:all" -- slices dealt round-robin,
// per-device streams and pools, root arrays per device, the gather, the combine in slice order -- run
// on it.  Nothing of this is part of the product: the shipped library has no device-aliasing path.
//
// Two layers are stood in for:
//   * the HIP runtime's device enumeration: hipGetDeviceCount reports k devices per physical GPU, and
//     hipSetDevice / hipGetDevice / hipGetDeviceProperties map device d to physical GPU d % real;
//   * RCCL: the real library refuses several ranks on one physical GPU ("Duplicate GPU detected"), so
//     ncclCommInitAll / ncclAllGather / ... are replaced by a same-process all-gather made of
//     device-to-device copies with event ordering between the ranks' streams.  The product looks RCCL
//     up with dlsym(RTLD_DEFAULT) first (csrc/comm_rccl.hpp), which finds these.  The real RCCL path is
//     exercised with one rank per real GPU by tests/test_gpu_comm.py.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

int factor()
{
    static const int k = [] {
        const char* e = getenv("VKMR_TEST_VIRTUAL_DEVICES");
        const int v = e ? atoi(e) : 1;
        return v < 1 ? 1 : (v > 16 ? 16 : v);
    }();
    return k;
}

template <typename F>
F next(const char* name)
{
    void* p = dlsym(RTLD_NEXT, name);
    if (!p) {
        fprintf(stderr, "virt_devices: %s not found behind the interposer\n", name);
        abort();
    }
    return reinterpret_cast<F>(p);
}

int real_count()
{
    static const int n = [] {
        int c = 0;
        if (next<hipError_t (*)(int*)>("hipGetDeviceCount")(&c) != hipSuccess) c = 0;
        return c;
    }();
    return n;
}

thread_local int t_virtual = 0;

}  // namespace

extern "C" {

hipError_t hipGetDeviceCount(int* count)
{
    const hipError_t e = next<hipError_t (*)(int*)>("hipGetDeviceCount")(count);
    if (e == hipSuccess) *count *= factor();
    return e;
}

hipError_t hipSetDevice(int dev)
{
    const int real = real_count();
    if (dev < 0 || real == 0 || dev >= real * factor()) return hipErrorInvalidDevice;
    t_virtual = dev;
    return next<hipError_t (*)(int)>("hipSetDevice")(dev % real);
}

hipError_t hipGetDevice(int* dev)
{
    *dev = t_virtual;
    return hipSuccess;
}

hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* prop, int dev)
{
    const int real = real_count();
    if (dev < 0 || real == 0 || dev >= real * factor()) return hipErrorInvalidDevice;
    return next<hipError_t (*)(hipDeviceProp_tR0600*, int)>("hipGetDevicePropertiesR0600")(prop, dev % real);
}

// ---- RCCL stand-in: every rank lives in this process, every buffer on the one physical GPU ----

struct FakeGroup {
    int nranks;
    struct Post { const void* send; void* recv; size_t bytes; hipStream_t stream; bool posted; };
    std::vector<Post> posts;
};
struct ncclComm {   // what ncclComm_t points to here
    FakeGroup* group;
    int rank;
    int dev;
};

static std::mutex g_mu;
static int g_depth = 0;
static std::vector<FakeGroup*> g_touched;

static ncclResult_t complete(FakeGroup* g)
{
    // rank r's data is ready once the work queued on its stream so far has run: one event per rank,
    // every receiving stream waits for all of them, then copies
    ncclResult_t res = ncclSuccess;
    auto ok = [&](hipError_t e) { if (e != hipSuccess) res = ncclUnhandledCudaError; };
    std::vector<hipEvent_t> ev((size_t)g->nranks);
    for (int r = 0; r < g->nranks; ++r) {
        ok(hipEventCreateWithFlags(&ev[(size_t)r], hipEventDisableTiming));
        ok(hipEventRecord(ev[(size_t)r], g->posts[(size_t)r].stream));
    }
    for (int j = 0; j < g->nranks; ++j) {
        for (int r = 0; r < g->nranks; ++r) ok(hipStreamWaitEvent(g->posts[(size_t)j].stream, ev[(size_t)r], 0));
        for (int r = 0; r < g->nranks; ++r)
            ok(hipMemcpyAsync(static_cast<char*>(g->posts[(size_t)j].recv) + (size_t)r * g->posts[(size_t)r].bytes, g->posts[(size_t)r].send,
                              g->posts[(size_t)r].bytes, hipMemcpyDeviceToDevice, g->posts[(size_t)j].stream));
    }
    for (auto e : ev) ok(hipEventDestroy(e));
    for (auto& p : g->posts) p.posted = false;
    return res;
}

ncclResult_t ncclGetVersion(int* v) { *v = 0; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL error (tests/c/virt_devices.cpp)"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int) { return ncclInvalidUsage; }   // one process only

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
    if (!comms || ndev < 1) return ncclInvalidArgument;
    FakeGroup* g = new FakeGroup;
    g->nranks = ndev;
    g->posts.assign((size_t)ndev, {nullptr, nullptr, 0, nullptr, false});
    for (int i = 0; i < ndev; ++i) comms[i] = new ncclComm{g, i, devlist ? devlist[i] : i};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    std::lock_guard<std::mutex> lock(g_mu);
    FakeGroup* g = c->group;
    delete c;
    if (--g->nranks == 0) delete g;   // the last rank takes the group with it
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_depth;
    return ncclSuccess;
}

static ncclResult_t flush()
{
    for (FakeGroup* g : g_touched) {
        for (const auto& p : g->posts)
            if (!p.posted) return ncclInvalidUsage;   // a rank missing from the group call
        const ncclResult_t r = complete(g);
        if (r != ncclSuccess) return r;
    }
    g_touched.clear();
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_depth == 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    return flush();
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t stream)
{
    if (!c || !send || !recv) return ncclInvalidArgument;
    const size_t width = (type == ncclUint32 || type == ncclInt32 || type == ncclFloat32) ? 4 : (type == ncclUint8 || type == ncclInt8) ? 1 : 0;
    if (width == 0) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lock(g_mu);
    FakeGroup* g = c->group;
    g->posts[(size_t)c->rank] = {send, recv, count * width, stream, true};
    bool seen = false;
    for (FakeGroup* t : g_touched) seen = seen || t == g;
    if (!seen) g_touched.push_back(g);
    if (g_depth == 0) return flush();   // outside a group only a one-rank communicator can complete
    return ncclSuccess;
}

}  // extern "C"
