// map_presort.hpp -- EXPERIMENTS BUILD ONLY (-DVKMR_EXPERIMENTS).  The map step as TWO kernels with no workgroup barrier in
// the hashing one (VKMR_MAP_VARIANT=20..22, profiles/r04_map_presort.txt):
//   map_sort_kernel          per tile of <= 1024 strings: the counting sort by block count of map_kernel's prologue, alone; the order
//                            (16-bit index inside the tile, longest first) goes to HBM: 8 bytes read + 2 written per string
//   map_hash_sorted_kernel   persistent wavefronts, each on its own: take a ticket (one of eight per-XCD counters; a drained XCD helps
//                            the next), read the 64 indices and metadata entries of that group, hash it with per-lane 16-byte loads,
//                            store the digests.  No LDS, no s_barrier, <= 64 VGPRs: 8 wavefronts per SIMD, none of them ever waiting
//                            for another -- map_kernel's wavefronts spend 16 % of a workgroup's life in its prologue and up to one
//                            group's time at its end waiting for the slowest of the eight.
// The price: per-lane loads re-read lines at the L2 (map_kernel's MODE 2 on short strings: 1.6x at the L2-fabric boundary) and the
// order crosses HBM twice.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../map_kernel.hpp"

#define VKMR_PRESORT_QUEUES 8
#define VKMR_PRESORT_QUEUE_STRIDE 1024   // words between two queue heads: 4 KiB, so that no two heads share a line, a channel or an atomic unit
                                         // (eight heads in one 32-byte run serialise like ONE word, ~88 tickets per us: 12 ms for 2^20 groups)

__device__ uint32_t g_presort_tickets[VKMR_PRESORT_QUEUES * VKMR_PRESORT_QUEUE_STRIDE];

template <int THREADS, int MAX_TILE>
__global__ __launch_bounds__(THREADS) void map_sort_kernel(const vkmr_metadata* __restrict__ meta, uint32_t count, uint64_t data_words, uint32_t tile,
                                                            uint16_t* __restrict__ order)
{
    __shared__ uint32_t s_hist[VKMR_MAP_BINS];
    __shared__ uint32_t s_binstart[VKMR_MAP_BINS];
    __shared__ uint16_t s_order[MAX_TILE];
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x == 0 && tid < VKMR_PRESORT_QUEUES) g_presort_tickets[tid * VKMR_PRESORT_QUEUE_STRIDE] = 0u;   // the hashing kernel's queues, for the launch behind this one
    const uint64_t tile_base = (uint64_t)blockIdx.x * tile;
    if (tile_base >= count) return;
    const uint32_t n_tile = (uint32_t)((count - tile_base < tile) ? count - tile_base : tile);
    if (tid < VKMR_MAP_BINS) s_hist[tid] = 0u;
    __syncthreads();
    constexpr int PER = MAX_TILE / THREADS;
    uint32_t key[PER], rank[PER];
    uint2 mdv[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = tid + k * THREADS;
        mdv[k] = make_uint2(0u, 0u);
        if (i < n_tile) mdv[k] = reinterpret_cast<const uint2*>(meta)[tile_base + i];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = tid + k * THREADS;
        key[k] = 0u; rank[k] = 0u;
        if (i < n_tile) {
            uint2 md = mdv[k];
            const unsigned long long avail = (md.x < data_words) ? (data_words - md.x) * 4ull : 0ull;   // as map_kernel: cut at the buffer's end
            md.y = (md.y > avail) ? (uint32_t)avail : md.y;
            const uint32_t nb = block_count(md.y);
            key[k] = nb < VKMR_MAP_BINS ? nb : (VKMR_MAP_BINS - 1u);
            rank[k] = atomicAdd(&s_hist[key[k]], 1u);
        }
    }
    __syncthreads();
    if (tid < VKMR_MAP_BINS) {
        uint32_t acc = 0u;
        for (uint32_t j = tid + 1u; j < VKMR_MAP_BINS; ++j) acc += s_hist[j];
        s_binstart[tid] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = tid + k * THREADS;
        if (i < n_tile) s_order[s_binstart[key[k]] + rank[k]] = (uint16_t)i;
    }
    __syncthreads();
    // coalesced copy out: two entries per lane
    for (uint32_t p = 2u * tid; p < n_tile; p += 2u * THREADS) {
        if (p + 1u < n_tile && ((tile_base + p) & 1ull) == 0ull) {
            *reinterpret_cast<uint32_t*>(order + tile_base + p) = (uint32_t)s_order[p] | ((uint32_t)s_order[p + 1u] << 16);
        } else {
            order[tile_base + p] = s_order[p];
            if (p + 1u < n_tile) order[tile_base + p + 1u] = s_order[p + 1u];
        }
    }
}

// The lane's index, recomputed where it is needed (two instructions) instead of a register held -- or spilled -- across the hash.
__device__ __forceinline__ uint32_t lane_id_now()
{
    uint32_t lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
}

// One group of 64 sorted strings of tile `t`, per-lane loads (map_kernel MODE 2's body).
__device__ __forceinline__ void hash_sorted_group(const uint32_t* __restrict__ data, uint64_t data_words, const vkmr_metadata* __restrict__ meta,
                                                  Node* __restrict__ out, uint64_t tile_base, uint32_t n_tile, uint32_t gi,
                                                  const uint16_t* __restrict__ order)
{
    const uint32_t lane = lane_id_now();
    const uint32_t pos = gi * 64u + lane;
    const bool has = pos < n_tile;
    const uint32_t id = has ? (uint32_t)order[tile_base + pos] : 0u;
    uint2 md = reinterpret_cast<const uint2*>(meta)[tile_base + id];
    const unsigned long long avail = (md.x < data_words) ? (data_words - md.x) * 4ull : 0ull;
    md.y = (md.y > avail) ? (uint32_t)avail : md.y;
    const uint32_t start = md.x, size = has ? md.y : 0u;
    const uint32_t nb = has ? block_count(size) : 0u;
    uint32_t H[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) H[i] = vkmr_dev::IV256[i];
    for (uint32_t b = 0; __any(b < nb); ++b) {
        uint32_t w[16];
        const uint64_t gbase = (uint64_t)start + ((uint64_t)b << 4);
        if (gbase + 16u <= data_words) {
            typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));
            const u32x4_u* src = reinterpret_cast<const u32x4_u*>(data + gbase);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4_u v = src[q];
                w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint64_t idx = gbase + i;
                w[i] = (idx < data_words) ? data[idx] : 0u;
            }
        }
        const uint64_t boff = (uint64_t)b << 6;
        const uint32_t r = (boff >= size) ? 0u : ((size - boff >= 64u) ? 64u : (uint32_t)(size - boff));
        uint32_t term = ((boff <= size) && (size - boff < 64u)) ? 0xFFFFFFFFu : 0u;
        asm("" : "+v"(term));
        const uint32_t kb = (r & 3u) << 3;
        const uint32_t keep = ~(0xFFFFFFFFu >> kb);
        const uint32_t padbit = 0x80000000u >> kb;
        const uint32_t full = r >> 2;
        uint32_t M[16];
        whole_word_masks(full, M);
        uint32_t prev = term;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t v = __builtin_bswap32(w[i]);
            const uint32_t bnd = __builtin_amdgcn_bitop3_b32(v, keep, padbit, 0xEA);
            const uint32_t u = bnd & prev;
            w[i] = __builtin_amdgcn_bitop3_b32(v, M[i], u, 0xE2);
            prev = M[i] & term;
        }
        if (b + 1u == nb) {
            w[14] = size >> 29;
            w[15] = size << 3;
        }
        if (b < nb) vkmr_dev::compress(H, w);
    }
    if (has) {
        uint32_t o[8];
        vkmr_dev::hash_digest(H, o);
        vkmr_dev::store_node(out + tile_base + id, o);
    }
}

// groups_per_queue: whole tiles per XCD queue; a (full) tile holds 2^gpt_log2 groups.
template <int WAVES, int TPG>   // TPG: consecutive groups per ticket
__global__ __launch_bounds__(WAVES * 64, 8) void map_hash_sorted_kernel(const uint32_t* __restrict__ data, uint64_t data_words,
                                                                         const vkmr_metadata* __restrict__ meta, uint32_t count, Node* __restrict__ out,
                                                                         uint32_t tile, const uint16_t* __restrict__ order, uint32_t gpt_log2,
                                                                         uint32_t groups_per_queue, uint32_t ngroups)
{
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    xcc &= (VKMR_PRESORT_QUEUES - 1u);
    for (uint32_t victim = 0; victim < VKMR_PRESORT_QUEUES; ++victim) {
        const uint32_t q = (xcc + victim) & (VKMR_PRESORT_QUEUES - 1u);
        const uint64_t lo = (uint64_t)q * groups_per_queue;
        if (lo >= ngroups) continue;
        const uint32_t gq = (uint32_t)((ngroups - lo < groups_per_queue) ? ngroups - lo : groups_per_queue);   // groups of this queue
        const uint32_t nq = (gq + TPG - 1) / TPG;                                                               // and its tickets
        uint32_t next = 0u;
        if (lane_id_now() == 0u) next = atomicAdd(&g_presort_tickets[q * VKMR_PRESORT_QUEUE_STRIDE], 1u);
        for (;;) {
            const uint32_t tk = __builtin_amdgcn_readfirstlane(next);
            if (tk >= nq) break;
            if (lane_id_now() == 0u) next = atomicAdd(&g_presort_tickets[q * VKMR_PRESORT_QUEUE_STRIDE], 1u);   // the ticket after this one: its round trip hides under the hashing
            for (uint32_t sub = 0; sub < TPG; ++sub) {
                if (tk * TPG + sub >= gq) break;
                const uint64_t g = lo + (uint64_t)tk * TPG + sub;
                const uint64_t t = g >> gpt_log2;                        // tiles hold a power of two of groups
                const uint32_t gi = (uint32_t)g & ((1u << gpt_log2) - 1u);
                const uint64_t tile_base = t * tile;
                if (tile_base >= count) break;
                const uint32_t n_tile = (uint32_t)((count - tile_base < tile) ? count - tile_base : tile);
                if (gi * 64u >= n_tile) continue;
                hash_sorted_group(data, data_words, meta, out, tile_base, n_tile, gi, order);
            }
        }
    }
}
