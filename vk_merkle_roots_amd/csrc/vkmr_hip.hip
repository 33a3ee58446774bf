// vkmr_hip.hip -- HIP kernels (gfx950) and the C ABI of include/vkmr_hip.h.
//
// Kernels (headers beside this file)
//   map_kernel.hpp      map_kernel             SHA-256d of every packed string      (SHA-256.comp:177-304)
//   reduce_kernels.hpp  reduce_pass_kernel     streaming sub-tree collapse per wave (SHA-256.comp:325-391)
//                       reduce_collapse_kernel / reduce_tail_kernel   the latency-bound top, __shfl_down
//                       reduce_level_kernel    one level per launch, cross-check    (SHA-256.comp:393-434)
//   sha256d_device.hpp  the SHA-256 round / compression building blocks
//   meta_kernels.hpp    sizes_*_kernel         metadata entries from 16-bit sizes   (Batches.cpp:64-121)
//
// Host side: plain launches on the caller's stream; no allocation, no sync inside
// the *_async entry points.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/vkmr_hip.h"
#include "sha256d_device.hpp"

using vkmr_dev::Node;

#include "map_kernel.hpp"
#include "meta_kernels.hpp"
#ifdef VKMR_EXPERIMENTS
#include "../../include/vkmr_hip_experiments.h"
#include "experiments/split_kernels.hpp"   // text -> packed batch on the device: measured, no gain, not shipped (include/vkmr_hip_experiments.h)
#endif
#include "reduce_kernels.hpp"
#include "reduce_plan.hpp"

// ============================================================================
// C ABI
// ============================================================================

static thread_local char g_err[512] = "";

static vkmr_status fail(vkmr_status code, const char* what, hipError_t e = hipSuccess)
{
    if (e != hipSuccess)
        snprintf(g_err, sizeof g_err, "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    else
        snprintf(g_err, sizeof g_err, "%s", what);
    return code;
}

static vkmr_status from_hip(hipError_t e, const char* what)
{
    if (e == hipSuccess) return VKMR_OK;
    (void)hipGetLastError();   // clear the sticky error
    if (e == hipErrorOutOfMemory) return fail(VKMR_ERR_OOM, what, e);
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return fail(VKMR_ERR_NO_DEVICE, what, e);
    return fail(VKMR_ERR_HIP, what, e);
}

#define VKMR_TRY(expr)                                   \
    do {                                                 \
        vkmr_status st__ = from_hip((expr), #expr);      \
        if (st__ != VKMR_OK) return st__;                \
    } while (0)

static inline hipStream_t S(vkmr_stream s) { return reinterpret_cast<hipStream_t>(s); }
static inline hipEvent_t E(vkmr_event e) { return reinterpret_cast<hipEvent_t>(e); }

extern "C" {

const char* vkmr_hip_last_error(void) { return g_err; }

#ifdef VKMR_STAMPS
// diagnostic build only: copies the stamp buffer out and clears it for the next launch
__attribute__((visibility("default"))) int vkmr_hip_debug_stamps(unsigned long long* out, int words)
{
    // The callers' streams are non-blocking ones: nothing orders them against the copy and the memset below (both on the
    // null stream) but these two device-wide waits.
    if (words < 0 || words > VKMR_STAMP_SLOTS * 8) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * words) != hipSuccess) return -1;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps)) != hipSuccess) return -1;
    if (hipMemset(p, 0, sizeof(unsigned long long) * VKMR_STAMP_SLOTS * 8) != hipSuccess) return -1;
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
#endif

const char* vkmr_hip_kernel_info(void);   // defined after the map entry point: it reports what the last launch chose

vkmr_status vkmr_hip_device_count(int* count)
{
    if (!count) return fail(VKMR_ERR_INVALID, "vkmr_hip_device_count: null out pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {   // no driver / no GPU: zero devices, like an empty Vulkan enumeration
        (void)hipGetLastError();
        *count = 0;
        fail(VKMR_OK, "hipGetDeviceCount", e);
        return VKMR_OK;
    }
    *count = n;
    return VKMR_OK;
}

vkmr_status vkmr_hip_device_name(int dev, char* buf, size_t buflen)
{
    if (!buf || buflen == 0) return fail(VKMR_ERR_INVALID, "vkmr_hip_device_name: null buffer");
    hipDeviceProp_t p;
    VKMR_TRY(hipGetDeviceProperties(&p, dev));
    // some ROCm installs leave the marketing name empty: fall back to the ISA name
    snprintf(buf, buflen, "%s", p.name[0] ? p.name : p.gcnArchName);
    return VKMR_OK;
}

vkmr_status vkmr_hip_device_mem_info(int dev, size_t* free_bytes, size_t* total_bytes)
{
    if (!free_bytes || !total_bytes) return fail(VKMR_ERR_INVALID, "vkmr_hip_device_mem_info: null out pointer");
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return VKMR_OK;
}

vkmr_status vkmr_hip_device_geometry(int dev, int* compute_units, int* wavefront)
{
    hipDeviceProp_t p;
    VKMR_TRY(hipGetDeviceProperties(&p, dev));
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (wavefront) *wavefront = p.warpSize;
    return VKMR_OK;
}

vkmr_status vkmr_hip_host_alloc(size_t bytes, void** out)
{
    if (!out || bytes == 0) return fail(VKMR_ERR_INVALID, "vkmr_hip_host_alloc: bad argument");
    void* p = nullptr;
    VKMR_TRY(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    memset(p, 0, bytes);
    *out = p;
    return VKMR_OK;
}

vkmr_status vkmr_hip_host_free(void* p)
{
    if (!p) return VKMR_OK;
    VKMR_TRY(hipHostFree(p));
    return VKMR_OK;
}

vkmr_status vkmr_hip_device_alloc(int dev, size_t bytes, void** out)
{
    if (!out || bytes == 0) return fail(VKMR_ERR_INVALID, "vkmr_hip_device_alloc: bad argument");
    VKMR_TRY(hipSetDevice(dev));
    void* p = nullptr;
    VKMR_TRY(hipMalloc(&p, bytes));
    *out = p;
    return VKMR_OK;
}

vkmr_status vkmr_hip_device_free(int dev, void* p)
{
    if (!p) return VKMR_OK;
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipFree(p));
    return VKMR_OK;
}

vkmr_status vkmr_hip_memset_async(int dev, vkmr_stream s, void* dst, int value, size_t bytes)
{
    if (!dst) return fail(VKMR_ERR_INVALID, "vkmr_hip_memset_async: null pointer");
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipMemsetAsync(dst, value, bytes, S(s)));
    return VKMR_OK;
}

vkmr_status vkmr_hip_memcpy_h2d_async(int dev, vkmr_stream s, void* dst_dev, const void* src_host, size_t bytes)
{
    if (!dst_dev || !src_host) return fail(VKMR_ERR_INVALID, "vkmr_hip_memcpy_h2d_async: null pointer");
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, S(s)));
    return VKMR_OK;
}

vkmr_status vkmr_hip_memcpy_d2h_async(int dev, vkmr_stream s, void* dst_host, const void* src_dev, size_t bytes)
{
    if (!dst_host || !src_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_memcpy_d2h_async: null pointer");
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, S(s)));
    return VKMR_OK;
}

vkmr_status vkmr_hip_stream_create(int dev, vkmr_stream* out)
{
    if (!out) return fail(VKMR_ERR_INVALID, "vkmr_hip_stream_create: null out pointer");
    VKMR_TRY(hipSetDevice(dev));
    hipStream_t s;
    VKMR_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = reinterpret_cast<vkmr_stream>(s);
    return VKMR_OK;
}

vkmr_status vkmr_hip_stream_destroy(int dev, vkmr_stream s)
{
    if (!s) return VKMR_OK;
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipStreamDestroy(S(s)));
    return VKMR_OK;
}

vkmr_status vkmr_hip_stream_sync(int dev, vkmr_stream s)
{
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipStreamSynchronize(S(s)));
    return VKMR_OK;
}

vkmr_status vkmr_hip_event_create(int dev, vkmr_event* out)
{
    if (!out) return fail(VKMR_ERR_INVALID, "vkmr_hip_event_create: null out pointer");
    VKMR_TRY(hipSetDevice(dev));
    hipEvent_t e;
    VKMR_TRY(hipEventCreate(&e));
    *out = reinterpret_cast<vkmr_event>(e);
    return VKMR_OK;
}

vkmr_status vkmr_hip_event_destroy(int dev, vkmr_event e)
{
    if (!e) return VKMR_OK;
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipEventDestroy(E(e)));
    return VKMR_OK;
}

vkmr_status vkmr_hip_event_record(int dev, vkmr_event e, vkmr_stream s)
{
    if (!e) return fail(VKMR_ERR_INVALID, "vkmr_hip_event_record: null event");
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipEventRecord(E(e), S(s)));
    return VKMR_OK;
}

vkmr_status vkmr_hip_event_query(int dev, vkmr_event e)
{
    if (!e) return fail(VKMR_ERR_INVALID, "vkmr_hip_event_query: null event");
    VKMR_TRY(hipSetDevice(dev));
    hipError_t r = hipEventQuery(E(e));
    if (r == hipErrorNotReady) {
        (void)hipGetLastError();
        return VKMR_NOT_READY;
    }
    return from_hip(r, "hipEventQuery");
}

vkmr_status vkmr_hip_event_wait(int dev, vkmr_event e)
{
    if (!e) return fail(VKMR_ERR_INVALID, "vkmr_hip_event_wait: null event");
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipEventSynchronize(E(e)));
    return VKMR_OK;
}

vkmr_status vkmr_hip_stream_wait_event(int dev, vkmr_stream s, vkmr_event e)
{
    if (!e) return fail(VKMR_ERR_INVALID, "vkmr_hip_stream_wait_event: null event");
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipStreamWaitEvent(S(s), E(e), 0));
    return VKMR_OK;
}

vkmr_status vkmr_hip_event_elapsed_ms(int dev, vkmr_event begin, vkmr_event end, float* ms)
{
    if (!begin || !end || !ms) return fail(VKMR_ERR_INVALID, "vkmr_hip_event_elapsed_ms: null argument");
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipEventElapsedTime(ms, E(begin), E(end)));
    return VKMR_OK;
}

// ---- map ------------------------------------------------------------------------

// What the last vkmr_hip_map_async of this process chose (reported by vkmr_hip_kernel_info).
enum { MAP_NONE = 0, MAP_STAGED, MAP_DIRECT512, MAP_DIRECT256, MAP_LONG512, MAP_LONG256, MAP_EXPERIMENT };
static std::atomic<int> g_last_map_mode{MAP_NONE};       // diagnostics only; several host threads may drive different devices
static std::atomic<uint32_t> g_last_map_tile{0};

// The shipped fetch modes (csrc/map_kernel.hpp).  The mode is chosen from the batch alone:
//   average packed string < 128 B   LDS-staged tiles (HBM traffic == algorithmic bytes)
//   128 B .. 512 B                  per-lane 16-byte loads, one 64-byte block per trip, 8 wavefronts per SIMD (256-lane workgroups when the launch is short)
//   512 B and more                  per-lane loads, TWO blocks (128 bytes) per trip: a 128-byte line is asked for by at most two trips instead
//                                   of three -- 8.6 instead of 11.3 GB at the L2-fabric boundary for 4.3 GB of rndm * 4096, the same
//                                   2.29 ms (87 VGPRs, 5 wavefronts per SIMD); at 150 B on average it is 3 % slower, hence the threshold
//                                   (profiles/r04_long_strings_two_blocks.txt)
// Round 2 also shipped a third mode for strings of 1 KiB and more -- whole 128-byte lines through a per-lane LDS window,
// 1.06x instead of 1.46x the algorithmic reads for 1-2 % of time.  Its 272 bytes of LDS per lane allow two wavefronts
// per SIMD, and since the issue pass (isa_prio_pass.py) the instruction pairing that decides the speed needs
// occupancy: 2.56 ms against 2.26 ms for the per-lane loads on rndm * 4096 (profiles/r03_long_strings_modes.txt).
// It stays in the experiments build (VKMR_MAP_VARIANT=5).
// (spelled with every template argument: the names must read exactly as a profiler prints them, provenance.py)
#define VKMR_MAP_STAGED_KERNEL map_kernel<512, 1024, 17664, 0, false, 0>
#define VKMR_MAP_DIRECT512_KERNEL map_kernel<512, 2048, 64, 2, true, 0>
#define VKMR_MAP_DIRECT256_KERNEL map_kernel<256, 2048, 64, 2, true, 0>
#define VKMR_MAP_LONG512_KERNEL map_kernel<512, 2048, 64, 5, true, 0>
#define VKMR_MAP_LONG256_KERNEL map_kernel<256, 2048, 64, 5, true, 0>

// Strings per LDS-staged tile: what is expected to fit the staging area, three standard deviations of a tile's
// size below it (string lengths spread like rndm's, uniform in [1, max]: sigma / mean of T strings is about
// 0.6 / sqrt(T)); a tile that overflows anyway falls back to per-lane loads inside the kernel.  The more strings a
// tile sorts the better: fewer of its groups straddle a block-count boundary, and 1024 strings are exactly two
// groups of 64 for each of the 8 wavefronts (profiles/r02_map_tile_fill.txt).  69 KiB of staging is what still
// lets two workgroups share a CU's LDS.  `fit_pct` (experiments build only) replaces the 3-sigma rule.
static uint32_t staged_tile(uint64_t data_words, uint32_t count, uint32_t max_tile, uint32_t stage_words, int fit_pct = 0)
{
    uint32_t tile = max_tile;
    if (data_words > 0) {
        const double r = (double)stage_words * (double)count / (double)data_words;   // strings that fill the area on average
        const double want = fit_pct ? r * fit_pct / 100.0 : r * (1.0 - 1.8 / __builtin_sqrt(r > 4.0 ? r : 4.0));
        const uint64_t fit = want > 0.0 ? (uint64_t)want : 0;
        if (fit >= max_tile / 4 && fit < tile) tile = (uint32_t)(fit & ~63ull);
    }
    // a launch too short to give every CU its two workgroups: smaller tiles, so that it still spreads over the chip
    const uint32_t spread = (uint32_t)((count / 512u) & ~63u);
    if (spread < tile) tile = spread < max_tile / 4 ? max_tile / 4 : spread;
    return tile;
}

// Tiles of the per-lane modes: up to 2048 strings, smaller when the batch is short so that it still spreads over
// the chip (>= ~1024 workgroups when it can).
static uint32_t direct_tile(uint32_t count)
{
    const uint32_t tile = (count / 1024u) & ~63u;
    return tile < 256u ? 256u : (tile > 2048u ? 2048u : tile);
}

static inline uint32_t tiles_of(uint32_t count, uint32_t tile) { return (uint32_t)(((uint64_t)count + tile - 1) / tile); }   // count + tile can pass 2^32

#ifdef VKMR_EXPERIMENTS
#include "map_experiments.hpp"   // tools build only: VKMR_MAP_VARIANT / _FIT / _TILE / _DYNLDS and the non-shipped instantiations
#endif

vkmr_status vkmr_hip_map_async(int dev, vkmr_stream s, const uint32_t* data_dev, uint64_t data_words,
                               const vkmr_metadata* meta_dev, uint32_t count, vkmr_digest* out_dev)
{
    if (count == 0) return VKMR_OK;
    if (!meta_dev || !out_dev || (!data_dev && data_words != 0))
        return fail(VKMR_ERR_INVALID, "vkmr_hip_map_async: null pointer");
    VKMR_TRY(hipSetDevice(dev));
    const uint64_t avg_words = (data_words + count - 1) / count;
    Node* out = reinterpret_cast<Node*>(out_dev);
#ifdef VKMR_EXPERIMENTS
    if (vkmr_map_experiment(S(s), data_dev, data_words, meta_dev, count, out, avg_words)) {
        g_last_map_mode = MAP_EXPERIMENT;
        VKMR_TRY(hipGetLastError());
        return VKMR_OK;
    }
#endif
    uint32_t tile = direct_tile(count);
    if (avg_words >= 32) {
        // strings of 128 B and more on average (a short launch: smaller workgroups spread it over the chip)
        if (avg_words >= 128 && tile >= 1024u) {
            g_last_map_mode = MAP_LONG512;
            hipLaunchKernelGGL((VKMR_MAP_LONG512_KERNEL), dim3(tiles_of(count, tile)), dim3(512), 0, S(s), data_dev, data_words, meta_dev, count, out, tile);
        } else if (avg_words >= 128) {
            g_last_map_mode = MAP_LONG256;
            hipLaunchKernelGGL((VKMR_MAP_LONG256_KERNEL), dim3(tiles_of(count, tile)), dim3(256), 0, S(s), data_dev, data_words, meta_dev, count, out, tile);
        } else if (tile >= 1024u) {
            g_last_map_mode = MAP_DIRECT512;
            hipLaunchKernelGGL((VKMR_MAP_DIRECT512_KERNEL), dim3(tiles_of(count, tile)), dim3(512), 0, S(s), data_dev, data_words, meta_dev, count, out, tile);
        } else {
            g_last_map_mode = MAP_DIRECT256;
            hipLaunchKernelGGL((VKMR_MAP_DIRECT256_KERNEL), dim3(tiles_of(count, tile)), dim3(256), 0, S(s), data_dev, data_words, meta_dev, count, out, tile);
        }
    } else {
        // short strings (a cache line holds several): the per-lane mode is 1-2 % faster but re-reads lines that
        // fell out of L2 (1.6x traffic, profiles/r01_map_fetch_modes.txt)
        tile = staged_tile(data_words, count, 1024, 17664);
        g_last_map_mode = MAP_STAGED;
        hipLaunchKernelGGL((VKMR_MAP_STAGED_KERNEL), dim3(tiles_of(count, tile)), dim3(512), 0, S(s), data_dev, data_words, meta_dev, count, out, tile);
    }
    g_last_map_tile = tile;
    VKMR_TRY(hipGetLastError());
    return VKMR_OK;
}

// ---- metadata from sizes (meta_kernels.hpp) ---------------------------------------------------------------------------
size_t vkmr_hip_sizes_scratch_bytes(uint32_t count)
{
    return ((size_t)(((uint64_t)count + VKMR_SIZES_BLOCK - 1) / VKMR_SIZES_BLOCK) + 1u) * sizeof(uint32_t);
}

vkmr_status vkmr_hip_metadata_from_sizes_async(int dev, vkmr_stream s, const uint16_t* sizes_dev, uint32_t count, uint32_t first_word,
                                               void* scratch_dev, vkmr_metadata* meta_dev)
{
    if (count == 0) return VKMR_OK;
    if (!sizes_dev || !scratch_dev || !meta_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_metadata_from_sizes_async: null pointer");
    if ((reinterpret_cast<uintptr_t>(sizes_dev) & 15u) || (reinterpret_cast<uintptr_t>(meta_dev) & 15u))
        return fail(VKMR_ERR_INVALID, "vkmr_hip_metadata_from_sizes_async: sizes and metadata must be 16-byte aligned");
    VKMR_TRY(hipSetDevice(dev));
    const uint32_t nblocks = (uint32_t)(((uint64_t)count + VKMR_SIZES_BLOCK - 1) / VKMR_SIZES_BLOCK);
    uint32_t* blocks = static_cast<uint32_t*>(scratch_dev);
    hipLaunchKernelGGL(sizes_block_words_kernel, dim3(nblocks), dim3(VKMR_SIZES_THREADS), 0, S(s), sizes_dev, count, blocks);
    hipLaunchKernelGGL(sizes_block_starts_kernel, dim3(1), dim3(VKMR_SIZES_THREADS), 0, S(s), blocks, nblocks, first_word);
    hipLaunchKernelGGL(sizes_expand_kernel, dim3(nblocks), dim3(VKMR_SIZES_THREADS), 0, S(s), sizes_dev, count, (const uint32_t*)blocks, meta_dev);
    VKMR_TRY(hipGetLastError());
    return VKMR_OK;
}

#ifdef VKMR_EXPERIMENTS
// ---- text -> packed batch (split_kernels.hpp) ---------------------------------------------------------------------------
namespace {
struct SplitScratch { uint32_t *blk_count, *blk_after, *wblocks; vkmr_split::Line* lines; size_t bytes; };
SplitScratch split_scratch(void* base, uint32_t text_bytes, uint32_t meta_capacity)
{
    auto up = [](size_t n) { return (n + 255u) & ~(size_t)255u; };
    const size_t nb = ((size_t)text_bytes + VKMR_SPLIT_BLOCK - 1) / VKMR_SPLIT_BLOCK + 1, nwb = ((size_t)meta_capacity + VKMR_SIZES_BLOCK - 1) / VKMR_SIZES_BLOCK + 1;
    char* p = static_cast<char*>(base);
    SplitScratch sc;
    sc.blk_count = reinterpret_cast<uint32_t*>(p); p += up(nb * 4);
    sc.blk_after = reinterpret_cast<uint32_t*>(p); p += up(nb * 4);
    sc.wblocks = reinterpret_cast<uint32_t*>(p); p += up(nwb * 4);
    sc.lines = reinterpret_cast<vkmr_split::Line*>(p); p += up((size_t)meta_capacity * sizeof(vkmr_split::Line));
    sc.bytes = (size_t)(p - static_cast<char*>(base));
    return sc;
}
}  // namespace

size_t vkmr_hip_split_scratch_bytes(uint32_t text_bytes, uint32_t meta_capacity) { return split_scratch(nullptr, text_bytes, meta_capacity).bytes; }

vkmr_status vkmr_hip_split_text_async(int dev, vkmr_stream s, const uint8_t* text_dev, uint32_t text_bytes, void* scratch_dev, uint32_t* data_dev,
                                      uint64_t data_capacity_words, vkmr_metadata* meta_dev, uint32_t meta_capacity, uint32_t* result_dev)
{
    if (!result_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_split_text_async: null result pointer");
    VKMR_TRY(hipSetDevice(dev));
    VKMR_TRY(hipMemsetAsync(result_dev, 0, 3 * sizeof(uint32_t), S(s)));
    if (text_bytes == 0) return VKMR_OK;
    if (!text_dev || !scratch_dev || !data_dev || !meta_dev || meta_capacity == 0)
        return fail(VKMR_ERR_INVALID, "vkmr_hip_split_text_async: null pointer");
    if ((reinterpret_cast<uintptr_t>(text_dev) & 15u) || text_bytes > 0xFFFFFFE0u)
        return fail(VKMR_ERR_INVALID, "vkmr_hip_split_text_async: the text must be 16-byte aligned and shorter than 4 GiB");
    const SplitScratch sc = split_scratch(scratch_dev, text_bytes, meta_capacity);
    const uint32_t nb = (uint32_t)(((uint64_t)text_bytes + VKMR_SPLIT_BLOCK - 1) / VKMR_SPLIT_BLOCK);
    const uint32_t nwb = (uint32_t)(((uint64_t)meta_capacity + VKMR_SIZES_BLOCK - 1) / VKMR_SIZES_BLOCK);
    hipLaunchKernelGGL(split_count_kernel, dim3(nb), dim3(VKMR_SPLIT_THREADS), 0, S(s), text_dev, text_bytes, sc.blk_count, sc.blk_after);
    hipLaunchKernelGGL(split_scan_kernel, dim3(1), dim3(VKMR_SPLIT_THREADS), 0, S(s), sc.blk_count, sc.blk_after, nb, result_dev);
    hipLaunchKernelGGL(split_lines_kernel, dim3(nb), dim3(VKMR_SPLIT_THREADS), 0, S(s), text_dev, text_bytes, (const uint32_t*)sc.blk_count,
                       (const uint32_t*)sc.blk_after, sc.lines, meta_capacity, result_dev);
    hipLaunchKernelGGL(split_block_words_kernel, dim3(nwb), dim3(VKMR_SIZES_THREADS), 0, S(s), (const vkmr_split::Line*)sc.lines, (const uint32_t*)result_dev,
                       meta_capacity, sc.wblocks);
    hipLaunchKernelGGL(sizes_block_starts_kernel, dim3(1), dim3(VKMR_SIZES_THREADS), 0, S(s), sc.wblocks, nwb, 0u);
    hipLaunchKernelGGL(split_expand_kernel, dim3(nwb), dim3(VKMR_SIZES_THREADS), 0, S(s), (const vkmr_split::Line*)sc.lines, result_dev, meta_capacity,
                       (const uint32_t*)sc.wblocks, meta_dev);
    hipLaunchKernelGGL(split_pack_kernel, dim3((meta_capacity + 255u) / 256u), dim3(256), 0, S(s), text_dev, (const vkmr_split::Line*)sc.lines,
                       (const vkmr_metadata*)meta_dev, result_dev, meta_capacity, data_dev, data_capacity_words);
    VKMR_TRY(hipGetLastError());
    return VKMR_OK;
}
#endif   // VKMR_EXPERIMENTS

// See the header: the first copy and the first launch of a process, taken out of the caller's pipeline.
vkmr_status vkmr_hip_warm_up(int dev, vkmr_stream s, unsigned what, size_t copy_bytes)
{
    VKMR_TRY(hipSetDevice(dev));
    if (!(what & (VKMR_WARM_KERNELS | VKMR_WARM_COPY))) return VKMR_OK;
    // one pinned and one device block, of 256 bytes at least: metadata {word 0, 4 bytes} at 0, the string's word at 64, its digest at 128
    const size_t bytes = (what & VKMR_WARM_COPY) && copy_bytes > 256 ? copy_bytes : 256;
    void *h = nullptr, *d = nullptr;
    hipError_t e = hipHostMalloc(&h, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&d, bytes);
    if (e == hipSuccess) {
        memset(h, 0, 256);
        static_cast<uint32_t*>(h)[1] = 4u;
        if (what & VKMR_WARM_COPY) e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, S(s));
        else e = hipMemsetAsync(d, 0, 256, S(s));
    }
    if (e == hipSuccess && (what & VKMR_WARM_KERNELS)) {
        char* base = static_cast<char*>(d);
        hipLaunchKernelGGL((VKMR_MAP_STAGED_KERNEL), dim3(1), dim3(512), 0, S(s), reinterpret_cast<const uint32_t*>(base + 64), (uint64_t)1,
                           reinterpret_cast<const vkmr_metadata*>(base), 1u, reinterpret_cast<Node*>(base + 128), 1024u);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(S(s));
    if (d) (void)hipFree(d);
    if (h) (void)hipHostFree(h);
    VKMR_TRY(e);
    return VKMR_OK;
}

#define VKMR_STR2(x) #x
#define VKMR_STR(x) VKMR_STR2(x)
const char* vkmr_hip_kernel_info(void)
{
    static thread_local char buf[512];
    const char* map = "map=(no launch yet; staged: " VKMR_STR((VKMR_MAP_STAGED_KERNEL)) ")";
    switch (g_last_map_mode.load()) {
        case MAP_STAGED: map = "map=" VKMR_STR((VKMR_MAP_STAGED_KERNEL)) " LDS-staged tiles sorted by block count"; break;
        case MAP_DIRECT512: map = "map=" VKMR_STR((VKMR_MAP_DIRECT512_KERNEL)) " per-lane 16-byte loads"; break;
        case MAP_DIRECT256: map = "map=" VKMR_STR((VKMR_MAP_DIRECT256_KERNEL)) " per-lane 16-byte loads, short launch"; break;
        case MAP_LONG512: map = "map=" VKMR_STR((VKMR_MAP_LONG512_KERNEL)) " per-lane 16-byte loads, two blocks per trip"; break;
        case MAP_LONG256: map = "map=" VKMR_STR((VKMR_MAP_LONG256_KERNEL)) " per-lane 16-byte loads, two blocks per trip, short launch"; break;
#ifdef VKMR_EXPERIMENTS
        case MAP_EXPERIMENT: map = "map=EXPERIMENT (VKMR_MAP_VARIANT; not a product build)"; break;
#endif
        default: break;
    }
#ifndef VKMR_BUILD_ID
#define VKMR_BUILD_ID "unknown"
#endif
    snprintf(buf, sizeof buf, "%s tile=%u reduce=reduce_pass_kernel(m<=%d)+reduce_collapse_kernel+reduce_tail_kernel(<=%d nodes) build=" VKMR_BUILD_ID, map,
             g_last_map_tile.load(), VKMR_PASS_MAXM, VKMR_TAIL_MAX);
    return buf;
}

// ---- reduce ---------------------------------------------------------------------

using vkmr_plan::ceil_shift;
using vkmr_plan::next_step;
typedef vkmr_plan::Step ReduceStep;
using vkmr_plan::STEP_BULK;
using vkmr_plan::STEP_COLLAPSE;
using vkmr_plan::STEP_TAIL;

// `height` levels must take `count` nodes to exactly one.  A tree over 2^64 leaves does not
// exist, so heights beyond 63 are refused rather than special-cased.
static bool height_ok(uint64_t count, uint32_t height)
{
    if (count == 0 || height > 63) return false;
    return ceil_shift(count, height) == 1;
}

size_t vkmr_hip_reduce_scratch_bytes(uint64_t count)
{
    // ping-pong: outputs of step 1 and step 2 (later steps are smaller), for ANY run of at
    // most `count` nodes -- see vkmr_plan::cells_upper_bound
    return (size_t)vkmr_plan::cells_upper_bound(count, 1) * sizeof(vkmr_digest);
}

// Reduces `nslices` slices (n_full nodes each, the last n_last) through `height`
// levels each; slice k's root goes to roots[k].  The step sequence is that of a full
// slice; a shorter last slice rides along (its surplus wavefronts exit at once).
static vkmr_status reduce_launch(hipStream_t stream, const Node* digests, uint32_t nslices, uint64_t n_full, uint64_t n_last,
                                 uint32_t height, Node* scratch, Node* roots, const ProofArgs* proofs = nullptr)
{
    const Node* in = digests;
    uint64_t n = n_full, nl = n_last, in_stride = n_full;
    uint32_t left = height;
    Node* bufA = scratch;
    Node* bufB = nullptr;
    const bool prove = proofs && proofs->k > 0;   // the kernels that also write the siblings of the proofs' path nodes (reduce_kernels.hpp)
    for (int pass = 0;; ++pass) {
        const ReduceStep st = next_step(n, left, nslices);
        const uint32_t level0 = height - left;     // tree level of this launch's input nodes
        SliceGeom g;
        g.n_full = n; g.n_last = nl; g.in_stride = in_stride; g.nslices = nslices;
        if (st.kind == STEP_TAIL) {
            g.out_stride = 1;
            if (prove) hipLaunchKernelGGL(reduce_tail_proofs_kernel, dim3(1, nslices), dim3(64), 0, stream, in, g, left, roots, *proofs, level0);
            else hipLaunchKernelGGL(reduce_tail_kernel, dim3(1, nslices), dim3(64), 0, stream, in, g, left, roots);
            VKMR_TRY(hipGetLastError());
            return VKMR_OK;
        }
        Node* out;
        if (pass == 0) {
            out = bufA;
            bufB = bufA + st.n_out * nslices;
        } else {
            out = (pass & 1) ? bufB : bufA;
        }
        g.out_stride = st.n_out;
        if (st.kind == STEP_BULK) {
            const uint32_t m = st.levels - 1u;
            const uint64_t waves = ceil_shift(n, 7 + m);
            const uint64_t grid = (waves + VKMR_PASS_WAVES - 1) / VKMR_PASS_WAVES;
            if (grid > 0x7fffffffull) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_async: slice too large");
            if (prove) hipLaunchKernelGGL(reduce_pass_proofs_kernel, dim3((uint32_t)grid, nslices), dim3(VKMR_PASS_WAVES * 64), 0, stream, in, g, out, m, *proofs, level0);
            else hipLaunchKernelGGL(reduce_pass_kernel, dim3((uint32_t)grid, nslices), dim3(VKMR_PASS_WAVES * 64), 0, stream, in, g, out, m);
        } else {
            const uint64_t cwaves = ceil_shift(n, 7);
            const dim3 cgrid((uint32_t)((cwaves + VKMR_COLLAPSE_WAVES - 1) / VKMR_COLLAPSE_WAVES), nslices);
            if (prove) hipLaunchKernelGGL(reduce_collapse_proofs_kernel, cgrid, dim3(VKMR_COLLAPSE_WAVES * 64), 0, stream, in, g, st.levels, out, *proofs, level0);
            else hipLaunchKernelGGL(reduce_collapse_kernel, cgrid, dim3(VKMR_COLLAPSE_WAVES * 64), 0, stream, in, g, st.levels, out);
        }
        VKMR_TRY(hipGetLastError());
        in = out;
        in_stride = st.n_out;
        n = st.n_out;
        nl = ceil_shift(nl, st.levels);
        left -= st.levels;
    }
}

vkmr_status vkmr_hip_reduce_async(int dev, vkmr_stream s, const vkmr_digest* digests_dev, uint64_t count,
                                  uint32_t height, void* scratch_dev, vkmr_digest* root_dev)
{
    if (!digests_dev || !root_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_async: null pointer");
    if (!height_ok(count, height))
        return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_async: height does not reduce count to one node");
    if (count > VKMR_TAIL_MAX && !scratch_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_async: null scratch");
    VKMR_TRY(hipSetDevice(dev));
    return reduce_launch(S(s), reinterpret_cast<const Node*>(digests_dev), 1, count, count, height,
                         reinterpret_cast<Node*>(scratch_dev), reinterpret_cast<Node*>(root_dev));
}

// ---- proof ----------------------------------------------------------------------

vkmr_status vkmr_hip_proof_async(int dev, vkmr_stream s, const vkmr_digest* digests_dev, uint64_t count, uint32_t height,
                                 uint64_t index, void* scratch_dev, vkmr_digest* siblings_dev, vkmr_digest* root_dev)
{
    if (!digests_dev || !siblings_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_proof_async: null pointer");
    if (!height_ok(count, height)) return fail(VKMR_ERR_INVALID, "vkmr_hip_proof_async: height does not reduce count to one node");
    if (index >= count) return fail(VKMR_ERR_INVALID, "vkmr_hip_proof_async: index out of range");
    if (count > VKMR_TAIL_MAX && !scratch_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_proof_async: null scratch");
    VKMR_TRY(hipSetDevice(dev));
    const Node* leaves = reinterpret_cast<const Node*>(digests_dev);
    Node* sib = reinterpret_cast<Node*>(siblings_dev);
    for (uint32_t l = 0; l < height; ++l) {
        // level l has cl nodes; the path node is p, its partner q (or p itself at the ragged right edge)
        const uint64_t cl = (l >= 64) ? 1 : ceil_shift(count, l);
        const uint64_t p = (l >= 64) ? 0 : (index >> l);
        uint64_t q = p ^ 1ull;
        if (q >= cl) q = p;
        // node q of level l = root of the sub-tree over leaves [q * 2^l, min((q + 1) * 2^l, count)), l levels
        const uint64_t lo = (l >= 64) ? 0 : (q << l);
        uint64_t n = (l >= 63) ? count - lo : ((count - lo < (1ull << l)) ? count - lo : (1ull << l));
        if (l == 0) {
            VKMR_TRY(hipMemcpyAsync(sib, leaves + lo, sizeof(Node), hipMemcpyDeviceToDevice, S(s)));
        } else {
            const vkmr_status st = reduce_launch(S(s), leaves + lo, 1, n, n, l, reinterpret_cast<Node*>(scratch_dev), sib + l);
            if (st != VKMR_OK) return st;
        }
    }
    if (root_dev)
        return reduce_launch(S(s), leaves, 1, count, count, height, reinterpret_cast<Node*>(scratch_dev), reinterpret_cast<Node*>(root_dev));
    return VKMR_OK;
}

// The reduction of vkmr_hip_reduce_async that ALSO writes the Merkle proofs of `k` leaves while it runs (the reference's to-do,
// README.md:118-120): siblings_dev[q * height + l] = the sibling of leaf indices[q]'s path node at level l.  Same launches, same
// root, no extra hash (reduce_kernels.hpp: note_siblings); vkmr_hip_proof_async stays as the independent cross-check.
vkmr_status vkmr_hip_reduce_proofs_async(int dev, vkmr_stream s, const vkmr_digest* digests_dev, uint64_t count, uint32_t height, void* scratch_dev,
                                         vkmr_digest* root_dev, const uint64_t* indices, uint32_t k, vkmr_digest* siblings_dev)
{
    if (k == 0) return vkmr_hip_reduce_async(dev, s, digests_dev, count, height, scratch_dev, root_dev);
    if (!digests_dev || !root_dev || !indices || !siblings_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_proofs_async: null pointer");
    if (k > VKMR_MAX_PROOFS) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_proofs_async: more than 16 proofs in one reduction");
    if (!height_ok(count, height)) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_proofs_async: height does not reduce count to one node");
    if (count > VKMR_TAIL_MAX && !scratch_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_proofs_async: null scratch");
    ProofArgs pa;
    pa.k = k;
    pa.height = height;
    pa.sib = reinterpret_cast<Node*>(siblings_dev);
    for (uint32_t q = 0; q < VKMR_MAX_PROOFS; ++q) pa.index[q] = 0;
    for (uint32_t q = 0; q < k; ++q) {
        if (indices[q] >= count) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_proofs_async: index out of range");
        pa.index[q] = indices[q];
    }
    VKMR_TRY(hipSetDevice(dev));
    if (height == 0) {   // one node, no level: the root is the node (reduce_tail_kernel), no sibling exists
        return reduce_launch(S(s), reinterpret_cast<const Node*>(digests_dev), 1, count, count, height, reinterpret_cast<Node*>(scratch_dev),
                             reinterpret_cast<Node*>(root_dev));
    }
    return reduce_launch(S(s), reinterpret_cast<const Node*>(digests_dev), 1, count, count, height, reinterpret_cast<Node*>(scratch_dev),
                         reinterpret_cast<Node*>(root_dev), &pa);
}

vkmr_status vkmr_hip_reduce_slices_async(int dev, vkmr_stream s, const vkmr_digest* digests_dev, uint32_t nslices,
                                         uint64_t capacity, uint64_t count_last, uint32_t height, void* scratch_dev,
                                         vkmr_digest* roots_dev)
{
    if (!digests_dev || !roots_dev || nslices == 0) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_slices_async: bad argument");
    if (count_last == 0 || count_last > capacity)
        return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_slices_async: bad slice geometry");
    if (!height_ok(nslices == 1 ? count_last : capacity, height))
        return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_slices_async: height does not reduce a slice to one node");
    if (capacity > VKMR_TAIL_MAX && !scratch_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_slices_async: null scratch");
    VKMR_TRY(hipSetDevice(dev));
    // grid.y carries the slice index: at most 32768 slices per launch sequence; longer runs go in
    // chunks on the same stream (the scratch is reused, the stream serialises them)
    const uint32_t chunk = 32768u;
    const Node* digests = reinterpret_cast<const Node*>(digests_dev);
    Node* roots = reinterpret_cast<Node*>(roots_dev);
    for (uint32_t first = 0; first < nslices; first += chunk) {
        const uint32_t n = (nslices - first < chunk) ? nslices - first : chunk;
        const bool has_last = (first + n == nslices);
        const uint64_t n_full = (nslices == 1) ? count_last : capacity;
        const vkmr_status st = reduce_launch(S(s), digests + (uint64_t)first * capacity, n, n_full, has_last ? count_last : capacity, height,
                                             reinterpret_cast<Node*>(scratch_dev), roots + first);
        if (st != VKMR_OK) return st;
    }
    return VKMR_OK;
}

size_t vkmr_hip_reduce_slices_scratch_bytes(uint64_t capacity, uint32_t nslices)
{
    if (nslices == 0) nslices = 1;
    // runs longer than 32768 slices are reduced in chunks (see vkmr_hip_reduce_slices_async): the
    // scratch must hold the largest chunk's passes -- the full chunks and the shorter last one
    const uint32_t full = nslices > 32768u ? 32768u : nslices;
    const uint32_t rest = nslices > 32768u ? nslices % 32768u : 0u;
    size_t cells = (size_t)vkmr_plan::cells_upper_bound(capacity, full) * full;
    if (rest) {
        const size_t c2 = (size_t)vkmr_plan::cells_upper_bound(capacity, rest) * rest;
        cells = c2 > cells ? c2 : cells;
    }
    return cells * sizeof(vkmr_digest);
}

size_t vkmr_hip_reduce_levels_scratch_bytes(uint64_t count)
{
    return (size_t)(ceil_shift(count, 1) + ceil_shift(count, 2) + 2) * sizeof(vkmr_digest);
}

vkmr_status vkmr_hip_reduce_levels_async(int dev, vkmr_stream s, const vkmr_digest* digests_dev, uint64_t count,
                                         uint32_t height, void* scratch_dev, vkmr_digest* root_dev)
{
    if (!digests_dev || !root_dev || !scratch_dev)
        return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_levels_async: null pointer");
    if (!height_ok(count, height))
        return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_levels_async: height does not reduce count to one node");
    VKMR_TRY(hipSetDevice(dev));
    const Node* in = reinterpret_cast<const Node*>(digests_dev);
    Node* bufA = reinterpret_cast<Node*>(scratch_dev);
    Node* bufB = bufA + ceil_shift(count, 1);
    uint64_t n = count;
    for (uint32_t lv = 0; lv < height; ++lv) {
        const uint64_t pairs = ceil_shift(n, 1);
        Node* out = (lv + 1 == height) ? reinterpret_cast<Node*>(root_dev) : ((lv & 1) ? bufB : bufA);
        const uint64_t grid = (pairs + 255) / 256;
        if (grid > 0x7fffffffull) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_levels_async: slice too large");
        hipLaunchKernelGGL(reduce_level_kernel, dim3((uint32_t)grid), dim3(256), 0, S(s), in, n, out);
        VKMR_TRY(hipGetLastError());
        in = out;
        n = pairs;
    }
    if (height == 0) VKMR_TRY(hipMemcpyAsync(root_dev, digests_dev, sizeof(vkmr_digest), hipMemcpyDeviceToDevice, S(s)));
    return VKMR_OK;
}

// ---- combine --------------------------------------------------------------------

static uint32_t combine_height(uint64_t n)
{
    uint32_t height = 1;   // at least one level: CpuSha256D::Root's do-while (SHA-256plus.cpp:515-547)
    while (ceil_shift(n, height) > 1) ++height;
    return height;
}

vkmr_status vkmr_hip_combine_async(int dev, vkmr_stream s, const vkmr_digest* roots_dev, uint32_t n, void* scratch_dev,
                                   vkmr_digest* root_dev)
{
    if (!roots_dev || !root_dev || n == 0) return fail(VKMR_ERR_INVALID, "vkmr_hip_combine_async: bad argument");
    return vkmr_hip_reduce_async(dev, s, roots_dev, n, combine_height(n), scratch_dev, root_dev);
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

#include "comm_rccl.hpp"
