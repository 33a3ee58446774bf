// meta_kernels.hpp -- metadata from sizes: the {first word, bytes} entry of every string of a batch whose strings lie back
// to back (string i + 1 starts on the word after string i) follows from the sizes alone,
//     start[i] = first_word + sum over j < i of ceil(size[j] / 4),
// which is what the reference's Batch::Push computes on the host while it appends (src/vkmr/Batches.cpp:64-121,
// WordCount :182-187).  A pipeline fed over PCIe sends 2 bytes per string (the 16-bit size) instead of the 8-byte entry --
// a tenth of what crosses the link for strings of 64 bytes -- and three short launches write the entries where the map
// kernel reads them.  Integer prefix sums: HBM-bound, 2 + 2 + 8 bytes per string.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vkmr_hip.h"

#define VKMR_SIZES_THREADS 256
#define VKMR_SIZES_PER 16                                        // sizes per lane: two 16-byte loads
#define VKMR_SIZES_BLOCK (VKMR_SIZES_THREADS * VKMR_SIZES_PER)   // strings per workgroup

namespace vkmr_sizes {

// This lane's 16 sizes (zero beyond `count`) and the words they take.
__device__ __forceinline__ uint32_t load16(const uint16_t* __restrict__ sizes, uint32_t count, uint32_t first, uint32_t (&sz)[VKMR_SIZES_PER])
{
    uint32_t words = 0;
    if (first + VKMR_SIZES_PER <= count) {
        const uint4 a = reinterpret_cast<const uint4*>(sizes + first)[0], b = reinterpret_cast<const uint4*>(sizes + first)[1];
        const uint32_t raw[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sz[2 * i] = raw[i] & 0xFFFFu;
            sz[2 * i + 1] = raw[i] >> 16;
        }
    } else {
#pragma unroll
        for (int i = 0; i < VKMR_SIZES_PER; ++i) sz[i] = (first + i < count) ? sizes[first + i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < VKMR_SIZES_PER; ++i) words += (sz[i] + 3u) >> 2;
    return words;
}

// Exclusive prefix of `v` over the workgroup's lanes (in lane order); *total = the workgroup's sum.
__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t* s_wave, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < VKMR_SIZES_THREADS / 64; ++w) {
        const uint32_t t = s_wave[w];
        before += (uint32_t)w < wave ? t : 0u;
        all += t;
    }
    *total = all;
    return before + incl - v;
}

}  // namespace vkmr_sizes

// 1. words per block of VKMR_SIZES_BLOCK strings
__global__ __launch_bounds__(VKMR_SIZES_THREADS) void sizes_block_words_kernel(const uint16_t* __restrict__ sizes, uint32_t count,
                                                                               uint32_t* __restrict__ block_words)
{
    __shared__ uint32_t s_wave[VKMR_SIZES_THREADS / 64];
    uint32_t sz[VKMR_SIZES_PER];
    const uint32_t first = blockIdx.x * VKMR_SIZES_BLOCK + threadIdx.x * VKMR_SIZES_PER;
    const uint32_t words = first < count ? vkmr_sizes::load16(sizes, count, first, sz) : 0u;
    uint32_t total;
    (void)vkmr_sizes::block_exclusive(words, s_wave, &total);
    if (threadIdx.x == 0) block_words[blockIdx.x] = total;
}

// 2. exclusive prefix over the blocks, in place, one workgroup (a batch of 2^20 strings has 256 blocks)
__global__ __launch_bounds__(VKMR_SIZES_THREADS) void sizes_block_starts_kernel(uint32_t* __restrict__ block_words, uint32_t nblocks, uint32_t first_word)
{
    __shared__ uint32_t s_wave[VKMR_SIZES_THREADS / 64];
    uint32_t carry = first_word;
    for (uint32_t base = 0; base < nblocks; base += VKMR_SIZES_THREADS) {   // wave-uniform trip count
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? block_words[i] : 0u;
        uint32_t total;
        const uint32_t ex = vkmr_sizes::block_exclusive(v, s_wave, &total);
        if (i < nblocks) block_words[i] = carry + ex;
        carry += total;
        __syncthreads();   // s_wave is reused by the next round
    }
}

// 3. the entries
__global__ __launch_bounds__(VKMR_SIZES_THREADS) void sizes_expand_kernel(const uint16_t* __restrict__ sizes, uint32_t count,
                                                                          const uint32_t* __restrict__ block_starts, vkmr_metadata* __restrict__ meta)
{
    __shared__ uint32_t s_wave[VKMR_SIZES_THREADS / 64];
    uint32_t sz[VKMR_SIZES_PER];
    const uint32_t first = blockIdx.x * VKMR_SIZES_BLOCK + threadIdx.x * VKMR_SIZES_PER;
    const uint32_t words = first < count ? vkmr_sizes::load16(sizes, count, first, sz) : 0u;
    uint32_t total;
    uint32_t w = block_starts[blockIdx.x] + vkmr_sizes::block_exclusive(words, s_wave, &total);
    if (first >= count) return;
    if (first + VKMR_SIZES_PER <= count) {
        uint4* out = reinterpret_cast<uint4*>(meta + first);   // two entries per 16-byte store
#pragma unroll
        for (int i = 0; i < VKMR_SIZES_PER; i += 2) {
            const uint32_t w1 = w + ((sz[i] + 3u) >> 2);
            out[i >> 1] = make_uint4(w, sz[i], w1, sz[i + 1]);
            w = w1 + ((sz[i + 1] + 3u) >> 2);
        }
    } else {
#pragma unroll
        for (int i = 0; i < VKMR_SIZES_PER; ++i) {
            if (first + i < count) {
                meta[first + i].start = w;
                meta[first + i].size = sz[i];
            }
            w += (sz[i] + 3u) >> 2;
        }
    }
}
