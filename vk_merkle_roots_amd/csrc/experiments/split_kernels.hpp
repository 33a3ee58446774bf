// split_kernels.hpp -- newline-separated text in device memory -> the packed batch layout (data words + metadata), on
// the device: what the reference's input loop and Batch::Push do on the host one string at a time (Input::Get,
// src/vkmr/Inputs.cpp:75-101: a line ends at '\n', '\r' is kept, an empty line is not a string; Batch::Push,
// src/vkmr/Batches.cpp:64-121: strings back to back on word boundaries, zero padding, {start word, size} per string).
// Five short passes, all HBM-bound integer work (text read three times, packed words written once):
//   1. per 16-byte piece: its newlines, and which of them end a non-empty line; per 4 KiB block: how many such ends, and
//      where its last newline is
//   2. one workgroup: prefix sums of the counts (-> the index of a block's first string), running maximum of the last
//      newlines (-> where the line that is open at the block's start began)
//   3. per piece again: {first byte, size} of every string, at its index
//   4. prefix sums of the strings' word counts (meta_kernels.hpp's scheme) -> {start word, size}
//   5. one lane per string: its bytes moved to its words (funnel shift by the byte offset), the last word's tail cleared
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/vkmr_hip.h"
#include "../meta_kernels.hpp"

#define VKMR_SPLIT_THREADS 256
#define VKMR_SPLIT_BLOCK (VKMR_SPLIT_THREADS * 16)   // text bytes per workgroup

namespace vkmr_split {

struct Line { uint32_t first, size; };   // a string's first byte in the text, its size in bytes

// The 16 bytes of piece `t`: bit i of *m = byte i is a newline, bit i of *e = it ends a non-empty line.  Bytes at and beyond
// `len` do not count.  The text starts "after a newline" (a leading '\n' is an empty line).
__device__ __forceinline__ void piece(const uint8_t* __restrict__ text, uint32_t len, uint32_t t, uint32_t* m, uint32_t* e)
{
    const uint32_t base = t * 16u;
    *m = 0u;
    *e = 0u;
    if (base >= len) return;
    const uint4 v = reinterpret_cast<const uint4*>(text)[t];   // the buffer is readable up to the next multiple of 16
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t mask = 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) mask |= (((w[i >> 2] >> (8 * (i & 3))) & 0xFFu) == 0x0Au ? 1u : 0u) << i;
    const uint32_t valid = (len - base >= 16u) ? 0xFFFFu : ((1u << (len - base)) - 1u);
    mask &= valid;
    const uint32_t before = (base == 0u) ? 1u : (text[base - 1u] == 0x0Au ? 1u : 0u);
    *m = mask;
    *e = mask & ~((mask << 1) | before) & 0xFFFFu;
}

// Exclusive prefix maximum of `v` over the workgroup's lanes (in lane order), seeded with `seed`; *total = the maximum over
// the seed and every lane.
__device__ __forceinline__ uint32_t block_exclusive_max(uint32_t v, uint32_t seed, uint32_t* s_wave, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl = o > incl ? o : incl;
    }
    uint32_t excl = __shfl_up(incl, 1);
    if (lane == 0u) excl = 0u;
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = seed, all = seed;
#pragma unroll
    for (int w = 0; w < VKMR_SPLIT_THREADS / 64; ++w) {
        const uint32_t t = s_wave[w];
        if ((uint32_t)w < wave) before = t > before ? t : before;
        all = t > all ? t : all;
    }
    *total = all;
    return excl > before ? excl : before;
}

}  // namespace vkmr_split

// 1. per block: strings that end in it; one past its last newline (0: it has none)
__global__ __launch_bounds__(VKMR_SPLIT_THREADS) void split_count_kernel(const uint8_t* __restrict__ text, uint32_t len, uint32_t* __restrict__ blk_count,
                                                                         uint32_t* __restrict__ blk_after)
{
    __shared__ uint32_t s_wave[VKMR_SPLIT_THREADS / 64];
    const uint32_t t = blockIdx.x * VKMR_SPLIT_THREADS + threadIdx.x;
    uint32_t m, e;
    vkmr_split::piece(text, len, t, &m, &e);
    const uint32_t after = m ? t * 16u + (32u - (uint32_t)__clz((int)m)) : 0u;   // one past the piece's last newline
    uint32_t total, top;
    (void)vkmr_sizes::block_exclusive((uint32_t)__popc(e), s_wave, &total);
    __syncthreads();
    (void)vkmr_split::block_exclusive_max(after, 0u, s_wave, &top);
    if (threadIdx.x == 0) {
        blk_count[blockIdx.x] = total;
        blk_after[blockIdx.x] = top;
    }
}

// 2. in place: blk_count -> index of the block's first string; blk_after -> where the line open at the block's start began.
//    result[0] = strings in all.
__global__ __launch_bounds__(VKMR_SPLIT_THREADS) void split_scan_kernel(uint32_t* __restrict__ blk_count, uint32_t* __restrict__ blk_after, uint32_t nblocks,
                                                                        uint32_t* __restrict__ result)
{
    __shared__ uint32_t s_wave[VKMR_SPLIT_THREADS / 64];
    uint32_t carry = 0u, open_at = 0u;
    for (uint32_t base = 0; base < nblocks; base += VKMR_SPLIT_THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t c = i < nblocks ? blk_count[i] : 0u, a = i < nblocks ? blk_after[i] : 0u;
        uint32_t total, top;
        const uint32_t ex = vkmr_sizes::block_exclusive(c, s_wave, &total);
        __syncthreads();
        const uint32_t mx = vkmr_split::block_exclusive_max(a, open_at, s_wave, &top);
        if (i < nblocks) {
            blk_count[i] = carry + ex;
            blk_after[i] = mx;
        }
        carry += total;
        open_at = top;
        __syncthreads();
    }
    if (threadIdx.x == 0) result[0] = carry;
}

// 3. {first byte, size} of every string
__global__ __launch_bounds__(VKMR_SPLIT_THREADS) void split_lines_kernel(const uint8_t* __restrict__ text, uint32_t len, const uint32_t* __restrict__ blk_first,
                                                                         const uint32_t* __restrict__ blk_open, vkmr_split::Line* __restrict__ lines,
                                                                         uint32_t capacity, uint32_t* __restrict__ result)
{
    __shared__ uint32_t s_wave[VKMR_SPLIT_THREADS / 64];
    const uint32_t t = blockIdx.x * VKMR_SPLIT_THREADS + threadIdx.x;
    uint32_t m, e;
    vkmr_split::piece(text, len, t, &m, &e);
    const uint32_t after = m ? t * 16u + (32u - (uint32_t)__clz((int)m)) : 0u;
    uint32_t total, top;
    uint32_t k = blk_first[blockIdx.x] + vkmr_sizes::block_exclusive((uint32_t)__popc(e), s_wave, &total);
    __syncthreads();
    uint32_t open_at = vkmr_split::block_exclusive_max(after, blk_open[blockIdx.x], s_wave, &top);   // where the line open at this piece's start began
    while (m) {
        const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
        m &= m - 1u;
        const uint32_t at = t * 16u + bit;   // a newline
        if (at > open_at) {
            if (k < capacity) lines[k] = {open_at, at - open_at};
            else result[2] = 1u;            // more strings than the caller's metadata holds
            ++k;
        }
        open_at = at + 1u;
    }
}

// 4a. words per block of VKMR_SIZES_BLOCK strings (the strings' sizes come from `lines`)
__device__ __forceinline__ uint32_t lines_words16(const vkmr_split::Line* __restrict__ lines, uint32_t count, uint32_t first, uint32_t (&sz)[VKMR_SIZES_PER])
{
    uint32_t words = 0;
#pragma unroll
    for (int i = 0; i < VKMR_SIZES_PER; ++i) {
        sz[i] = (first + i < count) ? lines[first + i].size : 0u;
        words += (sz[i] + 3u) >> 2;
    }
    return words;
}

__global__ __launch_bounds__(VKMR_SIZES_THREADS) void split_block_words_kernel(const vkmr_split::Line* __restrict__ lines, const uint32_t* __restrict__ result,
                                                                               uint32_t capacity, uint32_t* __restrict__ block_words)
{
    __shared__ uint32_t s_wave[VKMR_SIZES_THREADS / 64];
    const uint32_t count = result[0] < capacity ? result[0] : capacity;
    uint32_t sz[VKMR_SIZES_PER];
    const uint32_t first = blockIdx.x * VKMR_SIZES_BLOCK + threadIdx.x * VKMR_SIZES_PER;
    const uint32_t words = first < count ? lines_words16(lines, count, first, sz) : 0u;
    uint32_t total;
    (void)vkmr_sizes::block_exclusive(words, s_wave, &total);
    if (threadIdx.x == 0) block_words[blockIdx.x] = total;
}

// 4c. the entries (4b is sizes_block_starts_kernel); the last string's lane reports the words in all: result[1]
__global__ __launch_bounds__(VKMR_SIZES_THREADS) void split_expand_kernel(const vkmr_split::Line* __restrict__ lines, uint32_t* __restrict__ result, uint32_t capacity,
                                                                          const uint32_t* __restrict__ block_starts, vkmr_metadata* __restrict__ meta)
{
    __shared__ uint32_t s_wave[VKMR_SIZES_THREADS / 64];
    const uint32_t count = result[0] < capacity ? result[0] : capacity;
    uint32_t sz[VKMR_SIZES_PER];
    const uint32_t first = blockIdx.x * VKMR_SIZES_BLOCK + threadIdx.x * VKMR_SIZES_PER;
    const uint32_t words = first < count ? lines_words16(lines, count, first, sz) : 0u;
    uint32_t total;
    uint32_t w = block_starts[blockIdx.x] + vkmr_sizes::block_exclusive(words, s_wave, &total);
    if (first >= count) return;
#pragma unroll
    for (int i = 0; i < VKMR_SIZES_PER; ++i) {
        if (first + i < count) {
            meta[first + i].start = w;
            meta[first + i].size = sz[i];
        }
        w += (sz[i] + 3u) >> 2;
    }
    if (first + VKMR_SIZES_PER >= count) result[1] = w;   // this lane holds the last string
}

// 5. the bytes: one lane per string
__global__ __launch_bounds__(256) void split_pack_kernel(const uint8_t* __restrict__ text, const vkmr_split::Line* __restrict__ lines,
                                                         const vkmr_metadata* __restrict__ meta, uint32_t* __restrict__ result, uint32_t capacity,
                                                         uint32_t* __restrict__ data, uint64_t data_capacity_words)
{
    const uint32_t count = result[0] < capacity ? result[0] : capacity;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    if ((uint64_t)result[1] > data_capacity_words) {   // the packed words do not fit: nothing is written
        if (k == 0) result[2] = 1u;
        return;
    }
    const uint32_t first = lines[k].first, n = lines[k].size, nw = (n + 3u) >> 2;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(text) + (first >> 2);
    const uint32_t sh = (first & 3u) * 8u;
    uint32_t* dst = data + meta[k].start;
    uint32_t lo = src[0];
    for (uint32_t j = 0; j < nw; ++j) {
        // the word after the string's last may lie in the 16 bytes of slack behind the text: readable, and masked out below
        const uint32_t hi = sh ? src[j + 1u] : 0u;
        uint32_t v = sh ? __builtin_amdgcn_alignbit(hi, lo, sh) : lo;
        if (j + 1u == nw) v &= 0xFFFFFFFFu >> (8u * ((0u - n) & 3u));
        dst[j] = v;
        lo = sh ? hi : src[j + 1u < nw ? j + 1u : j];
    }
}
