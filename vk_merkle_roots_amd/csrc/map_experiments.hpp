// map_experiments.hpp -- EXPERIMENTS BUILD ONLY (-DVKMR_EXPERIMENTS: vk_merkle_roots_amd/build.py build_experiments(),
// build/ab/libexp.so).  The A/B knobs and the non-shipped map_kernel instantiations behind the measurements in
// profiles/ (r01_map_fetch_modes.txt, r02_map_tile_fill.txt, r02_long_strings_*.txt, r03_map_lds_schedule_ab.txt, r04_map_cycles_vs_power.txt).
// The product library is built without this file: its vkmr_hip_map_async picks the mode from the batch alone.
//   VKMR_MAP_VARIANT  which alternative (below); 0/unset = the shipped choice
//   VKMR_MAP_FIT      staged tiles: percent of the staging area to fill on average (50..100) instead of the 3-sigma rule
//   VKMR_MAP_TILE     per-lane / window modes: strings per tile (64..2048)
//   VKMR_MAP_DYNLDS   dynamic LDS bytes added to a launch: caps workgroups per CU
#pragma once
extern "C++" {   // this header is included from inside the ABI's extern "C" block
#include "experiments/map_presort.hpp"
#include "experiments/map_persist.hpp"

static int vkmr_exp_cus()
{
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t p;
        int dev = 0;
        (void)hipGetDevice(&dev);
        cus = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return cus;
}

// VKMR_MAP_VARIANT 20..22: sort kernel + barrier-free hashing kernel (experiments/map_presort.hpp).  The order array lives in a
// scratch buffer of this process (experiments only: the product's entry points allocate nothing).
template <int SORT_THREADS, int MAX_TILE, int WAVES, int TPG>
static void vkmr_map_presorted(hipStream_t s, const uint32_t* data_dev, uint64_t data_words, const vkmr_metadata* meta_dev, uint32_t count, Node* out,
                               uint32_t tile)
{
    static uint16_t* order = nullptr;
    static size_t order_cap = 0;
    const int cus = vkmr_exp_cus();
    if (order_cap < count) {
        if (order) (void)hipFree(order);
        order_cap = (size_t)count + 1024;
        if (hipMalloc(reinterpret_cast<void**>(&order), order_cap * sizeof(uint16_t)) != hipSuccess) { order = nullptr; order_cap = 0; return; }
    }
    const uint32_t tiles = tiles_of(count, tile), gpt = tile / 64u;
    const uint32_t tiles_per_queue = (tiles + VKMR_PRESORT_QUEUES - 1) / VKMR_PRESORT_QUEUES;
    hipLaunchKernelGGL((map_sort_kernel<SORT_THREADS, MAX_TILE>), dim3(tiles), dim3(SORT_THREADS), 0, s, meta_dev, count, data_words, tile, order);
    hipLaunchKernelGGL((map_hash_sorted_kernel<WAVES, TPG>), dim3((uint32_t)cus * (32u / WAVES)), dim3(WAVES * 64), 0, s, data_dev, data_words, meta_dev, count, out,
                       tile, (const uint16_t*)order, (uint32_t)__builtin_ctz(gpt), tiles_per_queue * gpt, tiles * gpt);
}
}   // extern "C++"

static bool vkmr_map_experiment(hipStream_t s, const uint32_t* data_dev, uint64_t data_words, const vkmr_metadata* meta_dev, uint32_t count,
                                Node* out, uint64_t avg_words)
{
    static const int variant = [] { const char* e = getenv("VKMR_MAP_VARIANT"); return e ? atoi(e) : 0; }();
    static const int fit_pct = [] { const char* e = getenv("VKMR_MAP_FIT"); const int v = e ? atoi(e) : 0; return (v < 50 || v > 100) ? 0 : v; }();
    static const int tile_override = [] { const char* e = getenv("VKMR_MAP_TILE"); return e ? atoi(e) : 0; }();
    static const int dyn_lds = [] { const char* e = getenv("VKMR_MAP_DYNLDS"); return e ? atoi(e) : 0; }();
    static const int stagger = [] { const char* e = getenv("VKMR_MAP_STAGGER"); return e ? atoi(e) : 5; }();   // variant 23: s_sleep(127) count of the late half
    if (variant == 0 && fit_pct == 0 && tile_override == 0 && dyn_lds == 0) return false;

    // (a lane takes ceil(max_tile / threads) metadata entries of a tile, so 640-string tiles on 512 lanes and 1024-string tiles on 768 work:
    // variants 26 and 27, six wavefronts per SIMD -- profiles/r04_map_cycles_vs_power.txt)
    auto launch_staged = [&](auto kern, uint32_t threads, uint32_t max_tile, uint32_t stage_words) {
        const uint32_t tile = staged_tile(data_words, count, max_tile, stage_words, fit_pct);
        hipLaunchKernelGGL(kern, dim3(tiles_of(count, tile)), dim3(threads), (size_t)dyn_lds, s, data_dev, data_words, meta_dev, count, out, tile);
    };
    uint32_t tile = direct_tile(count);
    if (tile_override >= 64 && tile_override <= 2048) tile = (uint32_t)tile_override & ~63u;
    const uint32_t grid = tiles_of(count, tile);
    auto launch_direct = [&](bool fullfast) {
        if (fullfast) {
            if (tile >= 1024u)
                hipLaunchKernelGGL((map_kernel<512, 2048, 64, 2, true>), dim3(grid), dim3(512), (size_t)dyn_lds, s, data_dev, data_words, meta_dev, count, out, tile);
            else
                hipLaunchKernelGGL((map_kernel<256, 2048, 64, 2, true>), dim3(grid), dim3(256), (size_t)dyn_lds, s, data_dev, data_words, meta_dev, count, out, tile);
        } else {
            if (tile >= 1024u)
                hipLaunchKernelGGL((map_kernel<512, 2048, 64, 2, false>), dim3(grid), dim3(512), 0, s, data_dev, data_words, meta_dev, count, out, tile);
            else
                hipLaunchKernelGGL((map_kernel<256, 2048, 64, 2, false>), dim3(grid), dim3(256), 0, s, data_dev, data_words, meta_dev, count, out, tile);
        }
    };
    auto launch_window = [&] {
        hipLaunchKernelGGL((map_kernel<512, 2048, 512 * VKMR_MAP_WIN_STRIDE, 4, true>), dim3(grid), dim3(512), (size_t)dyn_lds, s, data_dev, data_words,
                           meta_dev, count, out, tile);
    };
    switch (variant) {
        case 1: launch_staged(map_kernel<512, 1024, 16384, 0>, 512, 1024, 16384); break;     // LDS-staged tiles, 64 KiB (round 1's shipped shape)
        case 2: launch_staged(map_kernel<256, 512, 8192, 0>, 256, 512, 8192); break;         // LDS-staged tiles, 32 KiB
        case 3: hipLaunchKernelGGL((map_kernel<256, 2048, 5120, 1, true>), dim3(grid), dim3(256), 0, s, data_dev, data_words, meta_dev,
                                   count, out, tile); break;                                 // per-wavefront gather through LDS
        case 4: launch_direct(avg_words >= 32); break;                                       // per-lane 16-byte loads for every length (round 1's long-string mode)
        case 5: launch_window(); break;                                                      // line-aligned loads through a per-lane LDS window
        case 6: launch_staged(map_kernel<512, 1024, 17408, 0>, 512, 1024, 17408); break;     // LDS-staged tiles, 68 KiB
        case 7: launch_staged(map_kernel<512, 1024, 17664, 0>, 512, 1024, 17664); break;     // LDS-staged tiles, 69 KiB (the shipped shape, through the knobs)
        case 8: launch_staged(map_kernel<256, 1024, 17664, 0>, 256, 1024, 17664); break;     // the same tiles by 4 wavefronts instead of 8
        case 9: launch_staged(map_kernel<512, 1024, 17600, 0, false, 1>, 512, 1024, 17600); break;   // shipped shape, K[64] read from LDS
        case 10: launch_staged(map_kernel<512, 1024, 17600, 0, false, 2>, 512, 1024, 17600); break;  // K and the schedule ring in LDS (one workgroup per CU)
        case 11: launch_staged(map_kernel<512, 1024, 17600, 0, false, 0>, 512, 1024, 17600); break;  // control for 9/10: same staging, shipped compression
        case 12: launch_staged(map_kernel<1024, 1024, 17664, 0>, 1024, 1024, 17664); break;  // 16 wavefronts per tile: two workgroups = 8 wavefronts per SIMD
        case 13: launch_staged(map_kernel<1024, 2048, 34816, 0>, 1024, 2048, 34816); break;  // one 1024-lane workgroup per CU, 136 KiB tiles of 2048 strings
        case 26: launch_staged(map_kernel<512, 640, 11072, 0>, 512, 640, 11072); break;      // 640-string tiles of 51 KiB: THREE 8-wavefront workgroups per CU, 6 per SIMD
        case 27: launch_staged(map_kernel<768, 1024, 17664, 0>, 768, 1024, 17664); break;    // the shipped tile by 12 wavefronts: two workgroups = 6 per SIMD
        case 23: {   // persistent workgroups, the next tile in flight while this one is hashed (map_persist.hpp)
            const uint32_t tile = staged_tile(data_words, count, 1024, 17664, fit_pct), ntiles = tiles_of(count, tile);
            const uint32_t per_cu = dyn_lds ? 1u : 2u, want = (uint32_t)vkmr_exp_cus() * per_cu;
            hipLaunchKernelGGL((map_persist_kernel<512, 1024, 17664>), dim3(ntiles < want ? ntiles : want), dim3(512), (size_t)dyn_lds, s, data_dev, data_words,
                               meta_dev, count, out, tile, ntiles, (uint32_t)stagger);
            break;
        }
        case 24:   // per-lane loads, two blocks (128 bytes) per trip (map_kernel MODE 5)
            if (tile >= 1024u)
                hipLaunchKernelGGL((map_kernel<512, 2048, 64, 5, true>), dim3(grid), dim3(512), (size_t)dyn_lds, s, data_dev, data_words, meta_dev, count, out, tile);
            else
                hipLaunchKernelGGL((map_kernel<256, 2048, 64, 5, true>), dim3(grid), dim3(256), (size_t)dyn_lds, s, data_dev, data_words, meta_dev, count, out, tile);
            break;
        case 20: vkmr_map_presorted<256, 1024, 4, 1>(s, data_dev, data_words, meta_dev, count, out, 1024u); break;   // sort kernel + barrier-free hashing, tiles of 1024
        case 21: vkmr_map_presorted<256, 2048, 4, 1>(s, data_dev, data_words, meta_dev, count, out, 2048u); break;   // the same, tiles of 2048 (fewer mixed groups)
        case 22: vkmr_map_presorted<256, 1024, 4, 4>(s, data_dev, data_words, meta_dev, count, out, 1024u); break;   // four consecutive groups per ticket
        default:
            // the shipped choice, under the FIT / TILE / DYNLDS knobs
            if (avg_words >= 32) launch_direct(true);
            else launch_staged(map_kernel<512, 1024, 17664, 0>, 512, 1024, 17664);
            break;
    }
    return true;
}
