// fake_vkmr_hip.cpp -- TEST DOUBLE of the C ABI (include/vkmr_hip.h) for the host logic, no GPU.
//
// Built by tests/test_host_pipeline.py into tests/_build/fake/libvkmr_hip.so and put in front of the real
// library with LD_LIBRARY_PATH, so that the UNMODIFIED host code of the product -- the stream processor,
// batches, slices and their pools, mappings, reductions, the several-device combine (csrc/host/*.cpp, built
// with AddressSanitizer + UBSan for these tests) -- runs on a CPU-only machine and can be put under conditions a
// real GPU does not offer on demand:
//   VKMR_FAKE_DEVICES=k        k devices
//   VKMR_FAKE_HBM_BYTES=n      device_alloc fails with VKMR_ERR_OOM once a device would hold more than n bytes
//   VKMR_FAKE_EVENT_POLLS=p    an event reports VKMR_NOT_READY to its first p queries (asynchrony; default 2)
//   VKMR_FAKE_FAIL_EVENT=i     the i-th event completion (1-based, over all events) reports a device error
//   VKMR_FAKE_FAIL_REDUCE=i    the i-th vkmr_hip_reduce_async call fails at dispatch
//   VKMR_FAKE_NO_HASH=1        map and reduce write nothing (host-side timing of the pipeline only; roots are garbage)
// "Device" memory is host memory and every *_async call completes at once; hashing is done with the product's
// own "CPU" backend functions (csrc/host/cpu_sha256d.cpp), never with the oracle.  Scratch buffers are checked
// against the schedule of the real library (csrc/reduce_plan.hpp): a reduction whose scratch is smaller than
// what the real kernels would write is refused, so host-side sizing mistakes show up here too.
// This file is test infrastructure.  It is not a CPU fallback: nothing in the product loads it.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <vector>

#include "cpu_sha256d.hpp"
#include "reduce_plan.hpp"
#include "vkmr_hip.h"

namespace {

thread_local char g_err[256] = "";
std::mutex g_mu;

long env_long(const char* name, long def)
{
    const char* e = getenv(name);
    return e ? atol(e) : def;
}

int n_devices() { static const int n = (int)env_long("VKMR_FAKE_DEVICES", 1); return n < 0 ? 0 : (n > 64 ? 64 : n); }
size_t hbm_bytes() { static const long n = env_long("VKMR_FAKE_HBM_BYTES", 0); return n > 0 ? (size_t)n : 0; }

vkmr_status fail(vkmr_status code, const char* what)
{
    snprintf(g_err, sizeof g_err, "%s (fake ABI)", what);
    return code;
}

struct Allocation { int dev; size_t bytes; };
std::map<const void*, Allocation> g_allocs;      // device allocations by base pointer
std::vector<size_t> g_used(64, 0);

// bytes available from p to the end of the device allocation that contains it (0: not device memory)
size_t room_at(const void* p)
{
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_allocs.upper_bound(p);
    if (it == g_allocs.begin()) return 0;
    --it;
    const char* base = static_cast<const char*>(it->first);
    const char* q = static_cast<const char*>(p);
    if (q < base || q >= base + it->second.bytes) return 0;
    return (size_t)(base + it->second.bytes - q);
}

bool dev_ok(int dev) { return dev >= 0 && dev < n_devices(); }

struct Event { int polls_left = 0; bool recorded = false; bool failed = false; double at_ms = 0.0; };
long g_completions = 0, g_reduces = 0;

void do_reduce(const vkmr_digest* in, uint64_t count, uint32_t height, vkmr_digest* root)
{
    std::vector<uint32_t> nodes(8 * count);
    memcpy(nodes.data(), in, 32 * count);
    uint64_t n = count;
    for (uint32_t lv = 0; lv < height; ++lv) {
        const uint64_t pairs = (n + 1) / 2;
        for (uint64_t p = 0; p < pairs; ++p) {
            const uint32_t* l = nodes.data() + 16 * p;
            const uint32_t* r = (2 * p + 1 < n) ? l + 8 : l;
            uint32_t h[8];
            vkmr::cpu_sha256d_pair(l, r, h);
            memcpy(nodes.data() + 8 * p, h, 32);
        }
        n = pairs;
    }
    memcpy(root->data, nodes.data(), 32);
}

}  // namespace

struct vkmr_comm_s { std::vector<int> devs; };

extern "C" {

const char* vkmr_hip_last_error(void) { return g_err; }
const char* vkmr_hip_kernel_info(void) { return "fake ABI (tests/c/fake_vkmr_hip.cpp): host memory, CPU hashing"; }
const char* vkmr_hip_comm_info(void) { return "rccl=fake (tests/c/fake_vkmr_hip.cpp)"; }

vkmr_status vkmr_hip_device_count(int* count)
{
    if (!count) return fail(VKMR_ERR_INVALID, "device_count: null");
    *count = n_devices();
    return VKMR_OK;
}
vkmr_status vkmr_hip_device_name(int dev, char* buf, size_t n)
{
    if (!buf || !dev_ok(dev)) return fail(VKMR_ERR_NO_DEVICE, "device_name");
    snprintf(buf, n, "fake device %d", dev);
    return VKMR_OK;
}
vkmr_status vkmr_hip_device_mem_info(int dev, size_t* f, size_t* t)
{
    if (!dev_ok(dev) || !f || !t) return fail(VKMR_ERR_NO_DEVICE, "device_mem_info");
    *t = hbm_bytes() ? hbm_bytes() : ((size_t)1 << 40);
    *f = *t - g_used[(size_t)dev];
    return VKMR_OK;
}
vkmr_status vkmr_hip_device_geometry(int dev, int* cus, int* wave)
{
    if (!dev_ok(dev)) return fail(VKMR_ERR_NO_DEVICE, "device_geometry");
    if (cus) *cus = 256;
    if (wave) *wave = 64;
    return VKMR_OK;
}

vkmr_status vkmr_hip_host_alloc(size_t bytes, void** out)
{
    if (!out || bytes == 0) return fail(VKMR_ERR_INVALID, "host_alloc");
    *out = calloc(1, bytes);
    return *out ? VKMR_OK : fail(VKMR_ERR_OOM, "host_alloc");
}
vkmr_status vkmr_hip_host_free(void* p) { free(p); return VKMR_OK; }

vkmr_status vkmr_hip_device_alloc(int dev, size_t bytes, void** out)
{
    if (!out || bytes == 0) return fail(VKMR_ERR_INVALID, "device_alloc");
    if (!dev_ok(dev)) return fail(VKMR_ERR_NO_DEVICE, "device_alloc");
    std::lock_guard<std::mutex> lock(g_mu);
    if (hbm_bytes() && g_used[(size_t)dev] + bytes > hbm_bytes()) return fail(VKMR_ERR_OOM, "device_alloc: out of (fake) HBM");
    void* p = malloc(bytes);
    if (!p) return fail(VKMR_ERR_OOM, "device_alloc");
    memset(p, 0xCD, bytes);   // device memory is not zeroed
    g_allocs[p] = {dev, bytes};
    g_used[(size_t)dev] += bytes;
    *out = p;
    return VKMR_OK;
}
vkmr_status vkmr_hip_device_free(int dev, void* p)
{
    (void)dev;
    if (!p) return VKMR_OK;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_allocs.find(p);
    if (it == g_allocs.end()) return fail(VKMR_ERR_INVALID, "device_free: not a device allocation");
    g_used[(size_t)it->second.dev] -= it->second.bytes;
    g_allocs.erase(it);
    free(p);
    return VKMR_OK;
}

vkmr_status vkmr_hip_memset_async(int, vkmr_stream, void* dst, int v, size_t n)
{
    if (room_at(dst) < n) return fail(VKMR_ERR_INVALID, "memset: outside device memory");
    memset(dst, v, n);
    return VKMR_OK;
}
vkmr_status vkmr_hip_memcpy_h2d_async(int, vkmr_stream, void* dst, const void* src, size_t n)
{
    if (!dst || !src) return fail(VKMR_ERR_INVALID, "h2d: null");
    if (room_at(dst) < n) return fail(VKMR_ERR_INVALID, "h2d: destination outside device memory");
    memcpy(dst, src, n);
    return VKMR_OK;
}
vkmr_status vkmr_hip_memcpy_d2h_async(int, vkmr_stream, void* dst, const void* src, size_t n)
{
    if (!dst || !src) return fail(VKMR_ERR_INVALID, "d2h: null");
    if (room_at(src) < n) return fail(VKMR_ERR_INVALID, "d2h: source outside device memory");
    memcpy(dst, src, n);
    return VKMR_OK;
}

// VKMR_FAKE_COUNT_FORMS=1: at exit, how many map launches found their entries written by vkmr_hip_metadata_from_sizes_async
// (the batch crossed as 16-bit sizes) and how many did not (it crossed as entries)
static std::atomic<unsigned long> g_from_sizes{0}, g_maps{0};
static const vkmr_metadata* g_last_expanded = nullptr;
static std::atomic<unsigned long> g_splits{0};
static void report_forms()
{
    fprintf(stderr, "fake: batches described by sizes %lu, by entries %lu\n", g_from_sizes.load(), g_maps.load() - g_from_sizes.load());
    fprintf(stderr, "fake: texts split on the device %lu\n", g_splits.load());
}
size_t vkmr_hip_sizes_scratch_bytes(uint32_t count) { return ((size_t)count / 4096u + 2u) * 4u; }
vkmr_status vkmr_hip_metadata_from_sizes_async(int dev, vkmr_stream, const uint16_t* sizes, uint32_t count, uint32_t first_word, void* scratch,
                                               vkmr_metadata* meta)
{
    if (!dev_ok(dev) || (count && (!sizes || !scratch || !meta))) return fail(VKMR_ERR_INVALID, "metadata_from_sizes");
    if (count && (room_at(sizes) < (size_t)count * 2 || room_at(meta) < (size_t)count * sizeof(vkmr_metadata) || room_at(scratch) < vkmr_hip_sizes_scratch_bytes(count)))
        return fail(VKMR_ERR_INVALID, "metadata_from_sizes: outside device memory");
    uint32_t w = first_word;
    for (uint32_t i = 0; i < count; ++i) {
        meta[i].start = w;
        meta[i].size = sizes[i];
        w += (sizes[i] + 3u) / 4u;
    }
    g_last_expanded = meta;
    return VKMR_OK;
}

size_t vkmr_hip_split_scratch_bytes(uint32_t text_bytes, uint32_t meta_capacity) { return (size_t)text_bytes / 512u + (size_t)meta_capacity * 8u + 4096u; }
vkmr_status vkmr_hip_split_text_async(int dev, vkmr_stream, const uint8_t* text, uint32_t text_bytes, void* scratch, uint32_t* data, uint64_t data_capacity_words,
                                      vkmr_metadata* meta, uint32_t meta_capacity, uint32_t* result)
{
    if (!dev_ok(dev) || !result || room_at(result) < 12) return fail(VKMR_ERR_INVALID, "split_text");
    result[0] = result[1] = result[2] = 0;
    ++g_splits;
    if (text_bytes == 0) return VKMR_OK;
    if (!text || !scratch || !data || !meta || room_at(text) < text_bytes || room_at(meta) < (size_t)meta_capacity * 8 || room_at(data) < data_capacity_words * 4)
        return fail(VKMR_ERR_INVALID, "split_text: a buffer is not (large enough) device memory");
    uint64_t w = 0;
    uint32_t k = 0, open_at = 0;
    for (uint32_t i = 0; i < text_bytes; ++i) {
        if (text[i] != '\n') continue;
        const uint32_t n = i - open_at;
        if (n) {
            const uint64_t nw = (n + 3u) / 4u;
            if (k >= meta_capacity || w + nw > data_capacity_words) { result[2] = 1; return VKMR_OK; }
            meta[k].start = (uint32_t)w;
            meta[k].size = n;
            data[w + nw - 1] = 0;
            memcpy(data + w, text + open_at, n);
            w += nw;
            ++k;
        }
        open_at = i + 1;
    }
    result[0] = k;
    result[1] = (uint32_t)w;
    return VKMR_OK;
}

vkmr_status vkmr_hip_warm_up(int dev, vkmr_stream, unsigned, size_t) { return dev_ok(dev) ? VKMR_OK : fail(VKMR_ERR_INVALID, "warm_up"); }

vkmr_status vkmr_hip_stream_create(int dev, vkmr_stream* out)
{
    if (!out || !dev_ok(dev)) return fail(VKMR_ERR_INVALID, "stream_create");
    *out = reinterpret_cast<vkmr_stream>(new int(dev));
    return VKMR_OK;
}
vkmr_status vkmr_hip_stream_destroy(int, vkmr_stream s) { delete reinterpret_cast<int*>(s); return VKMR_OK; }
vkmr_status vkmr_hip_stream_sync(int, vkmr_stream) { return VKMR_OK; }

vkmr_status vkmr_hip_event_create(int dev, vkmr_event* out)
{
    if (!out || !dev_ok(dev)) return fail(VKMR_ERR_INVALID, "event_create");
    *out = reinterpret_cast<vkmr_event>(new Event);
    return VKMR_OK;
}
vkmr_status vkmr_hip_event_destroy(int, vkmr_event e) { delete reinterpret_cast<Event*>(e); return VKMR_OK; }
vkmr_status vkmr_hip_event_record(int, vkmr_event e, vkmr_stream)
{
    if (!e) return fail(VKMR_ERR_INVALID, "event_record");
    Event* ev = reinterpret_cast<Event*>(e);
    ev->recorded = true;
    ev->failed = false;
    ev->at_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    ev->polls_left = (int)env_long("VKMR_FAKE_EVENT_POLLS", 2);
    return VKMR_OK;
}
static vkmr_status complete(Event* ev)
{
    if (ev->polls_left >= 0) {   // first completion of this recording
        ev->polls_left = -1;
        std::lock_guard<std::mutex> lock(g_mu);
        if (++g_completions == env_long("VKMR_FAKE_FAIL_EVENT", 0)) ev->failed = true;
    }
    return ev->failed ? fail(VKMR_ERR_HIP, "event: injected device error") : VKMR_OK;
}
vkmr_status vkmr_hip_event_query(int, vkmr_event e)
{
    if (!e) return fail(VKMR_ERR_INVALID, "event_query");
    Event* ev = reinterpret_cast<Event*>(e);
    if (ev->polls_left > 0) {
        --ev->polls_left;
        return VKMR_NOT_READY;
    }
    return complete(ev);
}
vkmr_status vkmr_hip_event_wait(int, vkmr_event e)
{
    if (!e) return fail(VKMR_ERR_INVALID, "event_wait");
    Event* ev = reinterpret_cast<Event*>(e);
    if (ev->polls_left > 0) ev->polls_left = 0;
    return complete(ev);
}
vkmr_status vkmr_hip_stream_wait_event(int, vkmr_stream, vkmr_event) { return VKMR_OK; }
vkmr_status vkmr_hip_event_elapsed_ms(int, vkmr_event b, vkmr_event e, float* ms)
{
    if (!b || !e || !ms) return fail(VKMR_ERR_INVALID, "event_elapsed_ms");
    const double d = reinterpret_cast<Event*>(e)->at_ms - reinterpret_cast<Event*>(b)->at_ms;   // host time between the two records
    *ms = (float)(d > 1e-6 ? d : 1e-6);
    return VKMR_OK;
}

vkmr_status vkmr_hip_map_async(int dev, vkmr_stream, const uint32_t* data, uint64_t data_words, const vkmr_metadata* meta, uint32_t count,
                               vkmr_digest* out)
{
    if (count == 0) return VKMR_OK;
    if (!dev_ok(dev) || !meta || !out) return fail(VKMR_ERR_INVALID, "map: bad argument");
    if (room_at(meta) < (size_t)count * 8 || room_at(out) < (size_t)count * 32 || (data_words && room_at(data) < data_words * 4))
        return fail(VKMR_ERR_INVALID, "map: a buffer is not (large enough) device memory");
    static const bool no_hash = env_long("VKMR_FAKE_NO_HASH", 0) != 0;
    static const bool count_forms = env_long("VKMR_FAKE_COUNT_FORMS", 0) != 0 && atexit(report_forms) == 0;
    if (count_forms) {
        ++g_maps;
        if (g_last_expanded == meta) ++g_from_sizes;
        g_last_expanded = nullptr;
    }
    for (uint32_t i = 0; i < count && !no_hash; ++i) {
        uint64_t size = meta[i].size;
        const uint64_t avail = meta[i].start < data_words ? (data_words - meta[i].start) * 4 : 0;   // same cut as the kernel
        if (size > avail) size = avail;
        vkmr::cpu_sha256d_words(reinterpret_cast<const unsigned char*>(data + (avail ? meta[i].start : 0)), (size_t)size, out[i].data);
    }
    return VKMR_OK;
}

size_t vkmr_hip_reduce_scratch_bytes(uint64_t count) { return (size_t)vkmr_plan::cells_upper_bound(count, 1) * sizeof(vkmr_digest); }

vkmr_status vkmr_hip_reduce_async(int dev, vkmr_stream, const vkmr_digest* digests, uint64_t count, uint32_t height, void* scratch,
                                  vkmr_digest* root)
{
    if (!dev_ok(dev) || !digests || !root || count == 0 || height > 63 || vkmr_plan::ceil_shift(count, height) != 1)
        return fail(VKMR_ERR_INVALID, "reduce: bad argument");
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (++g_reduces == env_long("VKMR_FAKE_FAIL_REDUCE", 0)) return fail(VKMR_ERR_HIP, "reduce: injected dispatch failure");
    }
    if (room_at(digests) < count * 32 || room_at(root) < 32) return fail(VKMR_ERR_INVALID, "reduce: a buffer is not device memory");
    const uint64_t cells = vkmr_plan::cells_written(count, 1);   // what the real launch sequence writes
    if (cells && room_at(scratch) < cells * 32) return fail(VKMR_ERR_INVALID, "reduce: scratch smaller than the real kernels need");
    if (env_long("VKMR_FAKE_NO_HASH", 0) == 0) do_reduce(digests, count, height, root);
    return VKMR_OK;
}

vkmr_status vkmr_hip_combine_async(int dev, vkmr_stream s, const vkmr_digest* roots, uint32_t n, void* scratch, vkmr_digest* root)
{
    if (!roots || !root || n == 0) return fail(VKMR_ERR_INVALID, "combine: bad argument");
    uint32_t height = 1;
    while (vkmr_plan::ceil_shift(n, height) > 1) ++height;
    return vkmr_hip_reduce_async(dev, s, roots, n, height, scratch, root);
}

vkmr_status vkmr_hip_comm_init_all(const int* devs, int ndev, vkmr_comm* out)
{
    if (!devs || !out || ndev < 1) return fail(VKMR_ERR_INVALID, "comm_init_all");
    vkmr_comm_s* c = new vkmr_comm_s;
    for (int i = 0; i < ndev; ++i) {
        if (!dev_ok(devs[i])) { delete c; return fail(VKMR_ERR_NO_DEVICE, "comm_init_all"); }
        c->devs.push_back(devs[i]);
    }
    *out = c;
    return VKMR_OK;
}
vkmr_status vkmr_hip_comm_destroy(vkmr_comm c) { delete c; return VKMR_OK; }

vkmr_status vkmr_hip_gather_roots_async(vkmr_comm c, const vkmr_stream* streams, const vkmr_digest* const* roots, uint32_t per_rank,
                                        vkmr_digest* const* all)
{
    if (!c || !streams || !roots || !all || per_rank == 0) return fail(VKMR_ERR_INVALID, "gather: bad argument");
    const size_t n = c->devs.size();
    for (size_t i = 0; i < n; ++i)
        if (room_at(roots[i]) < (size_t)per_rank * 32 || room_at(all[i]) < n * per_rank * 32) return fail(VKMR_ERR_INVALID, "gather: buffer too small");
    for (size_t j = 0; j < n; ++j)
        for (size_t r = 0; r < n; ++r) memcpy(all[j] + r * per_rank, roots[r], (size_t)per_rank * 32);
    return VKMR_OK;
}

vkmr_status vkmr_hip_roots_in_slice_order_async(int, vkmr_stream, const vkmr_digest* gathered, uint32_t nranks, uint32_t per_rank, uint32_t total,
                                                vkmr_digest* out)
{
    if (!gathered || !out || nranks == 0 || per_rank == 0 || (uint64_t)total > (uint64_t)nranks * per_rank)
        return fail(VKMR_ERR_INVALID, "slice_order: bad argument");
    if (room_at(out) < (size_t)total * 32) return fail(VKMR_ERR_INVALID, "slice_order: output too small");
    for (uint32_t k = 0; k < total; ++k) out[k] = gathered[(size_t)(k % nranks) * per_rank + k / nranks];
    return VKMR_OK;
}

// ---- the rest of the header, so that the ctypes stub (which binds every declared symbol) and bench.py can run on the double ----

size_t vkmr_hip_reduce_slices_scratch_bytes(uint64_t capacity, uint32_t nslices)
{
    if (nslices == 0) nslices = 1;
    return (size_t)vkmr_plan::cells_upper_bound(capacity, nslices) * nslices * sizeof(vkmr_digest);
}

vkmr_status vkmr_hip_reduce_slices_async(int dev, vkmr_stream s, const vkmr_digest* digests, uint32_t nslices, uint64_t capacity,
                                         uint64_t count_last, uint32_t height, void* scratch, vkmr_digest* roots)
{
    if (!digests || !roots || nslices == 0 || count_last == 0 || count_last > capacity) return fail(VKMR_ERR_INVALID, "reduce_slices: bad argument");
    if (capacity > 128 && room_at(scratch) < vkmr_plan::cells_written(nslices == 1 ? count_last : capacity, nslices) * nslices * 32)
        return fail(VKMR_ERR_INVALID, "reduce_slices: scratch smaller than the real kernels need");
    for (uint32_t k = 0; k < nslices; ++k) {
        const uint64_t n = (k + 1 == nslices) ? count_last : capacity;
        if (vkmr_plan::ceil_shift(n, height) != 1) return fail(VKMR_ERR_INVALID, "reduce_slices: height");
        if (room_at(digests + (uint64_t)k * capacity) < n * 32 || room_at(roots + k) < 32) return fail(VKMR_ERR_INVALID, "reduce_slices: not device memory");
        do_reduce(digests + (uint64_t)k * capacity, n, height, roots + k);
    }
    (void)dev; (void)s;
    return VKMR_OK;
}

size_t vkmr_hip_reduce_levels_scratch_bytes(uint64_t count) { return (size_t)(vkmr_plan::ceil_shift(count, 1) + vkmr_plan::ceil_shift(count, 2) + 2) * 32; }
vkmr_status vkmr_hip_reduce_levels_async(int dev, vkmr_stream s, const vkmr_digest* digests, uint64_t count, uint32_t height, void* scratch,
                                         vkmr_digest* root)
{
    if (!scratch) return fail(VKMR_ERR_INVALID, "reduce_levels: null scratch");
    if (!dev_ok(dev) || !digests || !root || count == 0 || height > 63 || vkmr_plan::ceil_shift(count, height) != 1)
        return fail(VKMR_ERR_INVALID, "reduce_levels: bad argument");
    (void)s;
    do_reduce(digests, count, height, root);
    return VKMR_OK;
}

vkmr_status vkmr_hip_proof_async(int dev, vkmr_stream, const vkmr_digest* digests, uint64_t count, uint32_t height, uint64_t index, void* scratch,
                                 vkmr_digest* siblings, vkmr_digest* root)
{
    if (!dev_ok(dev) || !digests || !siblings || count == 0 || height > 63 || vkmr_plan::ceil_shift(count, height) != 1 || index >= count)
        return fail(VKMR_ERR_INVALID, "proof: bad argument");
    if (room_at(digests) < count * 32 || (height && room_at(siblings) < (size_t)height * 32)) return fail(VKMR_ERR_INVALID, "proof: not device memory");
    if (count > 128 && room_at(scratch) < 64) return fail(VKMR_ERR_INVALID, "proof: null scratch");
    std::vector<uint32_t> nodes(8 * count);   // level by level, the sibling of the path node picked at each
    memcpy(nodes.data(), digests, 32 * count);
    uint64_t n = count, p = index;
    for (uint32_t lv = 0; lv < height; ++lv) {
        const uint64_t q = ((p ^ 1ull) < n) ? (p ^ 1ull) : p;
        memcpy(siblings[lv].data, nodes.data() + 8 * q, 32);
        const uint64_t pairs = (n + 1) / 2;
        for (uint64_t k = 0; k < pairs; ++k) {
            const uint32_t* l = nodes.data() + 16 * k;
            const uint32_t* r = (2 * k + 1 < n) ? l + 8 : l;
            uint32_t h[8];
            vkmr::cpu_sha256d_pair(l, r, h);
            memcpy(nodes.data() + 8 * k, h, 32);
        }
        n = pairs;
        p >>= 1;
    }
    if (root) memcpy(root->data, nodes.data(), 32);
    return VKMR_OK;
}

vkmr_status vkmr_hip_reduce_proofs_async(int dev, vkmr_stream s, const vkmr_digest* digests, uint64_t count, uint32_t height, void* scratch,
                                         vkmr_digest* root, const uint64_t* indices, uint32_t k, vkmr_digest* siblings)
{
    if (k == 0) return vkmr_hip_reduce_async(dev, s, digests, count, height, scratch, root);
    if (!indices || !siblings || k > 16) return fail(VKMR_ERR_INVALID, "reduce_proofs: bad argument");
    if (room_at(siblings) < (size_t)k * height * 32) return fail(VKMR_ERR_INVALID, "reduce_proofs: siblings are not device memory (or too small)");
    const vkmr_status st = vkmr_hip_reduce_async(dev, s, digests, count, height, scratch, root);   // argument checks, injected failures, the root
    if (st != VKMR_OK) return st;
    std::vector<char> tmp(count > 128 ? 64 : 0);
    for (uint32_t q = 0; q < k; ++q) {
        if (indices[q] >= count) return fail(VKMR_ERR_INVALID, "reduce_proofs: index out of range");
        // same sibling definition as vkmr_hip_proof_async; the double computes it the slow way, level by level
        std::vector<uint32_t> nodes(8 * count);
        memcpy(nodes.data(), digests, 32 * count);
        uint64_t n = count, p = indices[q];
        for (uint32_t lv = 0; lv < height; ++lv) {
            const uint64_t o = ((p ^ 1ull) < n) ? (p ^ 1ull) : p;
            memcpy(siblings[(size_t)q * height + lv].data, nodes.data() + 8 * o, 32);
            const uint64_t pairs = (n + 1) / 2;
            for (uint64_t i = 0; i < pairs; ++i) {
                const uint32_t* l = nodes.data() + 16 * i;
                const uint32_t* r = (2 * i + 1 < n) ? l + 8 : l;
                uint32_t h[8];
                vkmr::cpu_sha256d_pair(l, r, h);
                memcpy(nodes.data() + 8 * i, h, 32);
            }
            n = pairs;
            p >>= 1;
        }
    }
    return VKMR_OK;
}

vkmr_status vkmr_hip_comm_create_id(void* id)
{
    if (!id) return fail(VKMR_ERR_INVALID, "comm_create_id");
    memset(id, 0x5A, VKMR_COMM_ID_BYTES);
    return VKMR_OK;
}
vkmr_status vkmr_hip_comm_init_rank(int dev, const void* id, int nranks, int rank, vkmr_comm* out)
{
    if (!id || !out || nranks != 1 || rank != 0 || !dev_ok(dev)) return fail(VKMR_ERR_COMM, "comm_init_rank: the fake ABI has no cross-process transport");
    vkmr_comm_s* c = new vkmr_comm_s;
    c->devs.push_back(dev);
    *out = c;
    return VKMR_OK;
}
vkmr_status vkmr_hip_comm_size(vkmr_comm c, int* nranks, int* nlocal)
{
    if (!c) return fail(VKMR_ERR_INVALID, "comm_size");
    if (nranks) *nranks = (int)c->devs.size();
    if (nlocal) *nlocal = (int)c->devs.size();
    return VKMR_OK;
}

void vkmr_hip_digest_hex(const vkmr_digest* d, char* hex)
{
    static const char digits[] = "0123456789abcdef";
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 4; ++b) {
            const unsigned v = (d->data[i] >> (24 - 8 * b)) & 0xffu;
            hex[8 * i + 2 * b] = digits[v >> 4];
            hex[8 * i + 2 * b + 1] = digits[v & 15u];
        }
    hex[64] = 0;
}

}  // extern "C"
