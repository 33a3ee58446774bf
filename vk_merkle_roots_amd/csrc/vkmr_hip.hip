// vkmr_hip.hip -- HIP kernels (gfx950) and the C ABI of include/vkmr_hip.h.
//
// Kernels
//   map_kernel          SHA-256d of every packed string        (replaces src/shaders/SHA-256.comp:177-304)
//   reduce_pass_kernel  streaming sub-tree collapse per wave    (replaces SHA-256.comp:325-391)
//   reduce_tail_kernel  top of a slice's tree, __shfl_down      (replaces SHA-256.comp:325-391, last passes)
//   reduce_level_kernel one level per launch, cross-check       (replaces SHA-256.comp:393-434)
//
// Host side: plain launches on the caller's stream; no allocation, no sync inside
// the *_async entry points.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/vkmr_hip.h"
#include "sha256d_device.hpp"

using vkmr_dev::Node;

// ============================================================================
// MAP
// ============================================================================
//
// One workgroup maps one TILE of up to 2048 consecutive strings:
//   1. metadata -> LDS, block count per string, counting sort of the tile by block
//      count (longest first) so that the 64 lanes of a wavefront run the same number
//      of compressions -- one lane per string without the sort runs every wavefront
//      at the pace of its longest string;
//   2. wavefronts pull groups of 64 sorted strings and hash them block by block:
//      16 message words per lane, byte swap, 0x80 / zero / bit-length padding by masks
//      (no branches), 64 unrolled rounds with the schedule ring in VGPRs.
// How the 16 words reach the lane is the template parameter MODE (all three are kept,
// parity-tested and timed against each other: profiles/r01_map_fetch_modes.txt,
// profiles/r01_map_fetch_vs_tile.txt):
//   MODE 0  the tile's packed bytes are copied to LDS with coalesced 16-byte HBM loads and
//           lanes read LDS (the layout the north star describes).  Shipped for short
//           strings: every byte crosses the HBM interface exactly once.
//   MODE 2  four 16-byte loads per lane straight from HBM/L2 (strings are 4-byte aligned;
//           gfx950 takes dword-aligned dwordx4).  No LDS, 7 waves/SIMD.  Shipped for long
//           strings (>= 128 B on average); for short ones it is 1-2 % faster than MODE 0 but
//           a 128-byte line shared by strings of different block counts is used at different
//           times and re-fetched once it has left L2 (1.6x the algorithmic bytes at L2/fabric).
//   MODE 1  per-wavefront gather: 16 lanes read one string's 64 contiguous bytes, four
//           strings per load, transposed through LDS rows.  Kept as the measured alternative.
// The kernel is bound by VALU issue, not by bytes: the three modes are within 5 % of
// each other.
// Digest i lands in out[i] whatever the processing order.

#define VKMR_MAP_STAGE_PAD 32
#define VKMR_MAP_GATHER_STRIDE 20   // words per string row in the gather area: 80 B keeps ds_read_b128 conflict-free
#define VKMR_MAP_BINS 64

#ifdef VKMR_MAP_STAMPS
// Diagnostic build only (tools/map_stamps.py): per-phase shader-clock totals of the map
// kernel, accumulated by lane 0 of every workgroup.  Never compiled into the product.
__device__ unsigned long long g_map_stamps[32768 * 8];   // 8 words per workgroup, no atomics
#define VKMR_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define VKMR_STAMP(var)
#endif

__device__ __forceinline__ uint32_t block_count(uint32_t size) { return (uint32_t)(((uint64_t)size + 8u) >> 6) + 1u; }

// THREADS lanes per workgroup, tiles of at most MAX_TILE strings, STAGE_WORDS words of LDS staging.
// GATHER selects how a tile that is not staged reads HBM: per wavefront through LDS rows
// (the long-string kernel) or, in the short-string kernel where that is the rare
// exception, simply per lane.
// FULLFAST adds a wave-uniform fast path for blocks in which every string of the group
// still has 64 bytes (long strings); short-string batches are faster without the test.
template <int THREADS, int MAX_TILE, int STAGE_WORDS, int MODE, bool FULLFAST = false>
__global__ __launch_bounds__(THREADS) void map_kernel(const uint32_t* __restrict__ data, uint64_t data_words,
                                                      const vkmr_metadata* __restrict__ meta, uint32_t count,
                                                      Node* __restrict__ out, uint32_t tile)
{
    constexpr int VKMR_MAP_THREADS = THREADS, VKMR_MAP_MAX_TILE = MAX_TILE, VKMR_MAP_STAGE_WORDS = STAGE_WORDS;
    constexpr bool GATHER = (MODE == 1);   // MODE 0: stage tiles in LDS; 1: per-wavefront gather; 2: per-lane 16-byte loads
    static_assert(!GATHER || STAGE_WORDS >= (THREADS / 64) * 64 * VKMR_MAP_GATHER_STRIDE, "staging area must hold the gather rows");
    __shared__ uint4 s_stage4[(VKMR_MAP_STAGE_WORDS + VKMR_MAP_STAGE_PAD) / 4];
    __shared__ uint2 s_meta[VKMR_MAP_MAX_TILE];
    __shared__ uint16_t s_order[VKMR_MAP_MAX_TILE];
    __shared__ uint32_t s_hist[VKMR_MAP_BINS];
    __shared__ uint32_t s_binstart[VKMR_MAP_BINS];
    __shared__ unsigned long long s_lo, s_hi;
    __shared__ uint32_t s_next;
    uint32_t* s_stage = reinterpret_cast<uint32_t*>(s_stage4);

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint64_t tile_base = (uint64_t)blockIdx.x * tile;
    if (tile_base >= count) return;
    const uint32_t n_tile = (uint32_t)((count - tile_base < tile) ? count - tile_base : tile);

    VKMR_STAMP(t_begin);
    // The prologue (sort + staging) is a few hundred instructions; a freshly launched
    // workgroup is the youngest on its SIMDs and would otherwise be starved by the older
    // workgroups' hashing, holding its LDS and wave slots idle.  Raise its issue priority
    // until it starts hashing itself.
    __builtin_amdgcn_s_setprio(3);
    if (tid < VKMR_MAP_BINS) s_hist[tid] = 0u;
    if (tid == 0) { s_lo = ~0ull; s_hi = 0ull; s_next = 0u; }
    __syncthreads();

    // ---- 1. metadata, keys, extent of the tile's packed bytes -------------------------
    constexpr int PER = VKMR_MAP_MAX_TILE / VKMR_MAP_THREADS;
    uint32_t key[PER], rank[PER];
    uint2 mdv[PER];
    unsigned long long lo = ~0ull, hi = 0ull;
    // all metadata loads of this lane are issued before any is used (one HBM round trip)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = tid + k * VKMR_MAP_THREADS;
        mdv[k] = make_uint2(0u, 0u);
        if (i < n_tile) mdv[k] = reinterpret_cast<const uint2*>(meta)[tile_base + i];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = tid + k * VKMR_MAP_THREADS;
        key[k] = 0u; rank[k] = 0u;
        if (i < n_tile) {
            const uint2 md = mdv[k];
            s_meta[i] = md;
            const uint32_t nb = block_count(md.y);
            key[k] = nb < VKMR_MAP_BINS ? nb : (VKMR_MAP_BINS - 1u);
            rank[k] = atomicAdd(&s_hist[key[k]], 1u);
            const unsigned long long b = md.x, e = b + (((unsigned long long)md.y + 3ull) >> 2);
            lo = b < lo ? b : lo;
            hi = e > hi ? e : hi;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long ol = __shfl_xor(lo, d), oh = __shfl_xor(hi, d);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if (lane == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    __syncthreads();

    // ---- 2. bin starts, longest strings first -------------------------------------------
    if (tid < VKMR_MAP_BINS) {
        uint32_t acc = 0u;
        for (uint32_t j = tid + 1u; j < VKMR_MAP_BINS; ++j) acc += s_hist[j];
        s_binstart[tid] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = tid + k * VKMR_MAP_THREADS;
        if (i < n_tile) s_order[s_binstart[key[k]] + rank[k]] = (uint16_t)i;
    }

    VKMR_STAMP(t_sorted);
    // ---- 3. stage the tile's packed words (coalesced) -------------------------------------
    const unsigned long long t_lo = s_lo, t_hi = s_hi;
    const unsigned long long a0 = t_lo & ~3ull;                 // 16-byte aligned start
    const bool staged = (MODE == 0) && (t_hi >= t_lo) && (t_hi - a0 <= VKMR_MAP_STAGE_WORDS) && (t_hi <= data_words) &&
                        ((reinterpret_cast<uintptr_t>(data) & 15u) == 0u);
    const uint32_t span = staged ? (uint32_t)(t_hi - a0) : 0u;  // words staged
    if (staged) {
        // every lane issues all of its 16-byte loads, then all of its LDS stores
        const uint4* src4 = reinterpret_cast<const uint4*>(data + a0);
        constexpr int NV = (VKMR_MAP_STAGE_WORDS / 4 + VKMR_MAP_THREADS - 1) / VKMR_MAP_THREADS;
        uint4 v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const uint32_t w = (tid + q * VKMR_MAP_THREADS) * 4u;
            v[q] = make_uint4(0u, 0u, 0u, 0u);
            if (w < span) {
                if (a0 + w + 4u <= data_words) {
                    v[q] = src4[w >> 2];
                } else {   // last, partial vector of the buffer
                    v[q].x = data[a0 + w];
                    v[q].y = (a0 + w + 1u < data_words) ? data[a0 + w + 1u] : 0u;
                    v[q].z = (a0 + w + 2u < data_words) ? data[a0 + w + 2u] : 0u;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const uint32_t w = (tid + q * VKMR_MAP_THREADS) * 4u;
            if (w < span) s_stage4[w >> 2] = v[q];
        }
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    VKMR_STAMP(t_staged);

    // ---- 4. hash groups of 64 sorted strings ---------------------------------------------
    const uint32_t ngroups = (n_tile + 63u) >> 6;
    for (;;) {
        uint32_t g = 0u;
        if (lane == 0) g = atomicAdd(&s_next, 1u);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= ngroups) break;
        const uint32_t pos = g * 64u + lane;
        const bool has = pos < n_tile;
        const uint32_t id = has ? s_order[pos] : 0u;
        const uint2 md = s_meta[id];
        const uint32_t start = md.x, size = has ? md.y : 0u;
        const uint32_t nb = has ? block_count(size) : 0u;

        uint32_t H[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) H[i] = vkmr_dev::IV256[i];

        // Gather path (tile not staged): the wavefront fetches its 64 strings' blocks
        // cooperatively -- 16 lanes read the 64 contiguous bytes of one string's block, 4
        // strings per load instruction -- through this wavefront's private LDS rows, so
        // HBM/L2 see 64-byte segments instead of 64 scattered dwords per instruction.
        const uint32_t sub = lane >> 4, wi = lane & 15u;
        uint32_t* wl = s_stage + (tid >> 6) * (64u * VKMR_MAP_GATHER_STRIDE);
        uint32_t gstart[GATHER ? 16 : 1];
        if (GATHER && !staged) {
#pragma unroll
            for (int j = 0; j < 16; ++j) gstart[GATHER ? j : 0] = __shfl(start, 4 * j + (int)sub);
        }

        for (uint32_t b = 0; __any(b < nb); ++b) {
            uint32_t w[16];
            // raw words of this block (garbage beyond the string is masked below)
            if (staged) {
                uint32_t base = (uint32_t)(start - a0) + (b << 4);
                base = base < span ? base : span;
#pragma unroll
                for (int i = 0; i < 16; ++i) w[i] = s_stage[base + i];
            } else if (!GATHER) {
                // per lane, straight from HBM/L2: four 16-byte loads (strings are only 4-byte
                // aligned; gfx950 takes dword-aligned dwordx4), scalar loads at the buffer's end
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
            } else {
                uint32_t g[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint64_t idx = (uint64_t)gstart[GATHER ? j : 0] + ((uint64_t)b << 4) + wi;
                    g[j] = (idx < data_words) ? data[idx] : 0u;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) wl[(4 * j + sub) * VKMR_MAP_GATHER_STRIDE + wi] = g[j];
                const uint4* row = reinterpret_cast<const uint4*>(wl + lane * VKMR_MAP_GATHER_STRIDE);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint4 v = row[q];
                    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
                }
            }
            // valid bytes of the string inside this block: 0..64
            const uint64_t boff = (uint64_t)b << 6;
            const uint32_t r = (boff >= size) ? 0u : ((size - boff >= 64u) ? 64u : (uint32_t)(size - boff));
            if (FULLFAST && __all(r == 64u || b >= nb)) {
                // every string of the group still has 64 bytes here: plain byte swap
#pragma unroll
                for (int i = 0; i < 16; ++i) w[i] = __builtin_bswap32(w[i]);
            } else {
                const bool term_here = (boff <= size) && (size - boff < 64u);   // the 0x80 byte falls in this block
                const uint32_t bw = term_here ? (r >> 2) : 16u;                 // word holding the terminator
                const uint32_t kb = (r & 3u) << 3;
                const uint32_t keep = kb ? (0xFFFFFFFFu << (32u - kb)) : 0u;
                const uint32_t padbit = 0x80000000u >> kb;
                const uint32_t full = r >> 2;                                   // whole data words
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t v = __builtin_bswap32(w[i]);
                    const uint32_t bnd = (v & keep) | padbit;
                    w[i] = ((uint32_t)i < full) ? v : (((uint32_t)i == bw) ? bnd : 0u);
                }
            }
            if (b + 1u == nb) {   // last block carries the 64-bit bit length (CPU path, SHA-256plus.cpp:100-117)
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
#ifdef VKMR_MAP_STAMPS
    {
        unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long rt = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && blockIdx.x < 32768u) {   // wavefront 0 of each workgroup: its own phase boundaries
            unsigned long long* o = g_map_stamps + (size_t)blockIdx.x * 8;
            o[0] = t_begin; o[1] = t_sorted; o[2] = t_staged; o[3] = t_end; o[4] = rt;
        }
    }
#endif
}

// ============================================================================
// REDUCE
// ============================================================================

#define VKMR_PASS_WAVES 4   // waves per workgroup in reduce_pass_kernel
#define VKMR_PASS_MAXM 4    // a wave consumes up to 2^4 chunks of 128 nodes: 5 levels per pass

__device__ __forceinline__ uint64_t level_count(uint64_t n, unsigned k) { return (n + ((1ull << k) - 1ull)) >> k; }

// Each wave walks 2^m chunks of 128 consecutive nodes.  A chunk gives 64 level-1
// nodes (one per lane).  Two such results of equal level merge into 64 nodes of
// the next level: lanes 0..31 hash pairs of the earlier (pending) result, lanes
// 32..63 pairs of the later one, so every step keeps all 64 lanes busy.  Pending
// results wait in this wave's LDS region (64 nodes per level); the later half is
// fetched from its lanes' registers with __shfl (ds_bpermute_b32).  After the last
// chunk the wave holds 64 nodes of level m+1 and writes them out coalesced.
// Pairing rule at every level: a node without a right sibling is paired with itself
// (src/shaders/SHA-256.comp:337, :363).
// Several equal-capacity slices can be reduced by one launch: blockIdx.y picks the
// slice (input `in_stride` nodes apart, output `out_stride` apart); the last slice
// may hold fewer nodes (`n_last`) than the others (`n_full`).
struct SliceGeom { uint64_t n_full, n_last, in_stride, out_stride; uint32_t nslices; };

__device__ __forceinline__ uint64_t slice_count(const SliceGeom& g) { return (blockIdx.y + 1u == g.nslices) ? g.n_last : g.n_full; }

__global__ __launch_bounds__(VKMR_PASS_WAVES * 64) void reduce_pass_kernel(const Node* __restrict__ in0, SliceGeom geom,
                                                                           Node* __restrict__ out0, uint32_t m)
{
    const uint64_t n_in = slice_count(geom);
    const Node* __restrict__ in = in0 + blockIdx.y * geom.in_stride;
    Node* __restrict__ out = out0 + blockIdx.y * geom.out_stride;
    __shared__ uint4 pend_store[VKMR_PASS_WAVES * VKMR_PASS_MAXM * 64 * 2];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t gwave = (uint64_t)blockIdx.x * VKMR_PASS_WAVES + wave;
    const uint64_t base0 = gwave * (128ull << m);
    if (base0 >= n_in) return;   // wave-uniform; no workgroup barrier is used below
    Node* pend = reinterpret_cast<Node*>(pend_store) + wave * (VKMR_PASS_MAXM * 64);

    uint32_t X[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t chunks = 1u << m;
    for (uint32_t c = 0; c < chunks; ++c) {
        uint32_t cc = c;
        uint32_t k = 0;   // this step turns level-k nodes into level-(k+1) nodes
        for (;;) {
            // first input node covered by this step, and this lane's output index
            const uint64_t first = base0 + 128ull * ((uint64_t)c + 1ull - (1ull << k));
            if (first < n_in) {   // wave-uniform: otherwise nothing below is a real node
                const uint64_t j = (first >> (k + 1)) + lane;
                const uint64_t ck = level_count(n_in, k);
                uint32_t l[8], r[8];
                if (k == 0) {
                    if (2 * j < ck) {
                        const Node a = vkmr_dev::load_node(in + 2 * j);
                        const Node b = (2 * j + 1 < ck) ? vkmr_dev::load_node(in + 2 * j + 1) : a;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { l[i] = a.w[i]; r[i] = b.w[i]; }
                    }
                } else {
                    const uint32_t src = (2u * lane) & 63u;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        l[i] = __shfl(X[i], src);
                        r[i] = __shfl(X[i], src + 1u);
                    }
                    if (lane < 32u) {
                        const Node a = pend[(k - 1) * 64 + 2 * lane];
                        const Node b = pend[(k - 1) * 64 + 2 * lane + 1];
#pragma unroll
                        for (int i = 0; i < 8; ++i) { l[i] = a.w[i]; r[i] = b.w[i]; }
                    }
                    if (2 * j + 1 >= ck) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) r[i] = l[i];
                    }
                }
                if (2 * j < ck) vkmr_dev::hash_pair(l, r, X);
            }
            ++k;
            if (!(cc & 1u)) break;
            cc >>= 1;
        }
        if (c + 1u != chunks) {
            Node t;
#pragma unroll
            for (int i = 0; i < 8; ++i) t.w[i] = X[i];
            pend[(k - 1) * 64 + lane] = t;
        }
    }
    const uint64_t jo = (base0 >> (m + 1)) + lane;
    if (jo < level_count(n_in, m + 1)) vkmr_dev::store_node(out + jo, X);
}

// Top of the tree: up to VKMR_TAIL_MAX nodes, exactly `levels` levels, one
// workgroup.  Level 1 comes from a coalesced pair load; the next six levels stay
// inside each 64-lane wavefront with __shfl_down, exactly the shape of the
// reference's subgroupShuffleDown loop (src/shaders/SHA-256.comp:346-377); up to
// sixteen wave results then meet in LDS and one wave finishes with __shfl_down.
// Any levels left once a single node remains hash that node with itself
// ("keep iterating", README.md:94).
__device__ __forceinline__ void shuffle_collapse(uint32_t (&X)[8], uint64_t idx0, uint32_t lane, uint64_t n_in,
                                                 uint32_t& done, uint32_t levels, uint32_t steps)
{
    for (uint32_t t = 0; t < steps && done < levels; ++t) {
        const uint64_t cnt = level_count(n_in, done);   // nodes alive at the current level
        const uint64_t me = idx0 >> t;                   // this lane's node index at that level
        uint32_t r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = __shfl_down(X[i], 1u << t);
        if (me + 1 >= cnt) {
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = X[i];
        }
        if ((lane & ((2u << t) - 1u)) == 0u && me < cnt) {
            uint32_t o[8];
            vkmr_dev::hash_pair(X, r, o);
#pragma unroll
            for (int i = 0; i < 8; ++i) X[i] = o[i];
        }
        ++done;
    }
}

__global__ __launch_bounds__(1024) void reduce_tail_kernel(const Node* __restrict__ in0, SliceGeom geom, uint32_t levels,
                                                           Node* __restrict__ root0)
{
    const uint32_t n_in = (uint32_t)slice_count(geom);
    const Node* __restrict__ in = in0 + blockIdx.y * geom.in_stride;
    Node* __restrict__ root = root0 + blockIdx.y * geom.out_stride;
    __shared__ Node wave_out[16];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint32_t X[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t done = 0;
    if (levels == 0) {   // n_in == 1: the root is the node itself
        if (tid == 0) *root = in[0];
        return;
    }
    if (2 * tid < n_in) {
        const Node a = vkmr_dev::load_node(in + 2 * tid);
        const Node b = (2 * tid + 1 < n_in) ? vkmr_dev::load_node(in + 2 * tid + 1) : a;
        vkmr_dev::hash_pair(a.w, b.w, X);
    }
    done = 1;
    shuffle_collapse(X, tid, lane, n_in, done, levels, 6);
    if (done < levels) {   // uniform across the workgroup
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wave_out[wave].w[i] = X[i];
        }
        __syncthreads();
        if (wave == 0) {
            if (lane < 16u && lane < (blockDim.x >> 6)) {
#pragma unroll
                for (int i = 0; i < 8; ++i) X[i] = wave_out[lane].w[i];
            }
            shuffle_collapse(X, lane, lane, n_in, done, levels, 4);
            while (done < levels) {   // a single node left: pair it with itself
                uint32_t o[8];
                vkmr_dev::hash_pair(X, X, o);
#pragma unroll
                for (int i = 0; i < 8; ++i) X[i] = o[i];
                ++done;
            }
        }
    }
    if (tid == 0) vkmr_dev::store_node(root, X);
}

// Middle of the tree, where there are too few nodes to keep every SIMD busy: one
// wavefront per workgroup (so the wavefronts spread over all CUs) collapses 128
// nodes through `levels` (1..7) levels -- a coalesced pair load, then __shfl_down
// steps as in the reference's subgroup shader.  Lane utilisation is poor by
// construction here but these passes are latency-bound: what counts
// is the ~9 us one wavefront needs per level, not the idle lanes.
__global__ __launch_bounds__(64) void reduce_collapse_kernel(const Node* __restrict__ in0, SliceGeom geom, uint32_t levels,
                                                             Node* __restrict__ out0)
{
    const uint64_t n_in = slice_count(geom);
    const Node* __restrict__ in = in0 + blockIdx.y * geom.in_stride;
    Node* __restrict__ out = out0 + blockIdx.y * geom.out_stride;
    const uint32_t lane = threadIdx.x;
    const uint64_t j = (uint64_t)blockIdx.x * 64u + lane;   // level-1 node of this lane
    uint32_t X[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (2 * j < n_in) {
        const Node a = vkmr_dev::load_node(in + 2 * j);
        const Node b = (2 * j + 1 < n_in) ? vkmr_dev::load_node(in + 2 * j + 1) : a;
        vkmr_dev::hash_pair(a.w, b.w, X);
    }
    uint32_t done = 1;
    shuffle_collapse(X, j, lane, n_in, done, levels, 6);
    const uint64_t jo = j >> (levels - 1u);
    if ((lane & ((1u << (levels - 1u)) - 1u)) == 0u && jo < level_count(n_in, levels)) vkmr_dev::store_node(out + jo, X);
}

// One level, one lane per pair (reference's BasicReduction shader, SHA-256.comp:393-434,
// with `>=` bounds and self-pairing instead of the duplicate-last buffer copy,
// src/vkmr/Reductions.cpp:299-342).
__global__ __launch_bounds__(256) void reduce_level_kernel(const Node* __restrict__ in, uint64_t n_in, Node* __restrict__ out)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * p >= n_in) return;
    const Node a = vkmr_dev::load_node(in + 2 * p);
    const Node b = (2 * p + 1 < n_in) ? vkmr_dev::load_node(in + 2 * p + 1) : a;
    uint32_t o[8];
    vkmr_dev::hash_pair(a.w, b.w, o);
    vkmr_dev::store_node(out + p, o);
}

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

#ifdef VKMR_MAP_STAMPS
__attribute__((visibility("default"))) int vkmr_hip_debug_stamps(unsigned long long* out, int words)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_map_stamps), sizeof(unsigned long long) * words) == hipSuccess ? 0 : -1;
}
#endif

const char* vkmr_hip_kernel_info(void)
{
    return "map=map_kernel(tile-sorted by block count; LDS-staged tiles, per-lane dwordx4 for long strings) reduce=reduce_pass_kernel(m<=4)+reduce_collapse_kernel+reduce_tail_kernel";
}

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

vkmr_status vkmr_hip_map_async(int dev, vkmr_stream s, const uint32_t* data_dev, uint64_t data_words,
                               const vkmr_metadata* meta_dev, uint32_t count, vkmr_digest* out_dev)
{
    if (count == 0) return VKMR_OK;
    if (!meta_dev || !out_dev || (!data_dev && data_words != 0))
        return fail(VKMR_ERR_INVALID, "vkmr_hip_map_async: null pointer");
    VKMR_TRY(hipSetDevice(dev));
    // VKMR_MAP_VARIANT picks an alternative fetch mode / geometry for A/B timing; 0 = shipped.
    static const int variant = [] { const char* e = getenv("VKMR_MAP_VARIANT"); return e ? atoi(e) : 0; }();
    const uint64_t avg_words = (data_words + count - 1) / count;
    Node* out = reinterpret_cast<Node*>(out_dev);
    // staged mode: strings per tile = what is expected to fit the LDS staging area
    auto launch_staged = [&](auto kern, uint32_t threads, uint32_t max_tile, uint32_t stage_words) {
        uint32_t tile = max_tile;
        if (avg_words > 0) {
            const uint64_t fit = (uint64_t)(stage_words * 0.9) / avg_words;
            if (fit >= max_tile / 4 && fit < tile) tile = (uint32_t)(fit & ~63ull);
        }
        hipLaunchKernelGGL(kern, dim3((count + tile - 1) / tile), dim3(threads), 0, S(s), data_dev, data_words, meta_dev, count, out, tile);
    };
    // tiles of up to 2048 strings, smaller when the batch is short so that it still
    // spreads over the chip (>= ~1024 workgroups when it can)
    uint32_t tile = (count / 1024u) & ~63u;
    tile = tile < 256u ? 256u : (tile > 2048u ? 2048u : tile);
    static const int tile_override = [] { const char* e = getenv("VKMR_MAP_TILE"); return e ? atoi(e) : 0; }();   // experiments only
    if (tile_override >= 64 && tile_override <= 2048) tile = (uint32_t)tile_override & ~63u;
    const uint32_t grid = (count + tile - 1) / tile;
    auto launch_direct = [&](bool fullfast) {
        if (fullfast) {
            if (tile >= 1024u)
                hipLaunchKernelGGL((map_kernel<512, 2048, 64, 2, true>), dim3(grid), dim3(512), 0, S(s), data_dev, data_words, meta_dev, count, out, tile);
            else
                hipLaunchKernelGGL((map_kernel<256, 2048, 64, 2, true>), dim3(grid), dim3(256), 0, S(s), data_dev, data_words, meta_dev, count, out, tile);
        } else {
            if (tile >= 1024u)
                hipLaunchKernelGGL((map_kernel<512, 2048, 64, 2, false>), dim3(grid), dim3(512), 0, S(s), data_dev, data_words, meta_dev, count, out, tile);
            else
                hipLaunchKernelGGL((map_kernel<256, 2048, 64, 2, false>), dim3(grid), dim3(256), 0, S(s), data_dev, data_words, meta_dev, count, out, tile);
        }
    };
    switch (variant) {
        case 1: launch_staged(map_kernel<512, 1024, 16384, 0>, 512, 1024, 16384); break;     // LDS-staged tiles, 64 KiB
        case 2: launch_staged(map_kernel<256, 512, 8192, 0>, 256, 512, 8192); break;         // LDS-staged tiles, 32 KiB
        case 3: hipLaunchKernelGGL((map_kernel<256, 2048, 5120, 1, true>), dim3(grid), dim3(256), 0, S(s), data_dev, data_words, meta_dev,
                                   count, out, tile); break;                                 // per-wavefront gather through LDS
        case 4: launch_direct(avg_words >= 32); break;                                       // per-lane 16-byte loads for every length
        default:
            // Shipped: short strings (< 128 B on average: a cache line holds several) go through
            // LDS-staged tiles -- HBM traffic == algorithmic bytes; the per-lane mode is 1-2 %
            // faster but re-reads lines that fell out of L2 (1.6x traffic, profiles/
            // r01_map_fetch_modes.txt).  Long strings read per lane with the full-block fast path.
            if (avg_words >= 32)
                launch_direct(true);
            else
                launch_staged(map_kernel<512, 1024, 16384, 0>, 512, 1024, 16384);
            break;
    }
    VKMR_TRY(hipGetLastError());
    return VKMR_OK;
}

// ---- reduce ---------------------------------------------------------------------

static inline uint64_t ceil_shift(uint64_t n, unsigned k) { return (n + ((1ull << k) - 1ull)) >> k; }

// Levels a bulk pass collapses for n input nodes per slice: the largest m+1 (m <= MAXM)
// that still leaves enough wavefronts (over all slices of the launch) to fill 256 CUs.
static uint32_t pick_m(uint64_t n, uint32_t nslices)
{
    const uint64_t target_waves = 4096;
    for (int m = VKMR_PASS_MAXM; m > 0; --m)
        if (ceil_shift(n, 7 + m) * nslices >= target_waves && (128ull << m) <= n) return (uint32_t)m;
    return 0;
}

static bool height_ok(uint64_t count, uint32_t height)
{
    if (count == 0) return false;
    if (height >= 64) return true;
    return ceil_shift(count, height) == 1;
}

// One step of the reduction schedule for n nodes with `left` levels to go:
//   bulk     n/128 >= 2048 wavefronts: reduce_pass_kernel, m+1 levels, every lane busy
//   collapse 128 < n: reduce_collapse_kernel, up to 7 levels, one wavefront per CU slot
//   tail     n <= 128: reduce_tail_kernel, one wavefront, all remaining levels
struct ReduceStep { int kind; uint32_t levels; uint64_t n_out; };
enum { STEP_BULK = 0, STEP_COLLAPSE = 1, STEP_TAIL = 2 };

static ReduceStep next_step(uint64_t n, uint32_t left, uint32_t nslices)
{
    ReduceStep st;
    if (n <= 128) {
        st.kind = STEP_TAIL; st.levels = left; st.n_out = 1;
    } else if (ceil_shift(n, 7) * nslices >= 2048) {
        st.kind = STEP_BULK; st.levels = pick_m(n, nslices) + 1u; st.n_out = ceil_shift(n, st.levels);
    } else {
        st.kind = STEP_COLLAPSE; st.levels = 7; st.n_out = ceil_shift(n, 7);
    }
    return st;
}

// Scratch cells one slice needs when `nslices` slices are reduced together.
static uint64_t scratch_cells(uint64_t count, uint32_t nslices)
{
    uint64_t n = count, total = 0;
    for (int pass = 0; pass < 2 && n > 128; ++pass) {
        n = next_step(n, 64, nslices).n_out;
        total += n;
    }
    return total + 2;
}

size_t vkmr_hip_reduce_scratch_bytes(uint64_t count)
{
    // ping-pong: outputs of step 1 and step 2 (later steps are smaller)
    return (size_t)scratch_cells(count, 1) * sizeof(vkmr_digest);
}

// Reduces `nslices` slices (n_full nodes each, the last n_last) through `height`
// levels each; slice k's root goes to roots[k].  The step sequence is that of a full
// slice; a shorter last slice rides along (its surplus wavefronts exit at once).
static vkmr_status reduce_launch(hipStream_t stream, const Node* digests, uint32_t nslices, uint64_t n_full, uint64_t n_last,
                                 uint32_t height, Node* scratch, Node* roots)
{
    const Node* in = digests;
    uint64_t n = n_full, nl = n_last, in_stride = n_full;
    uint32_t left = height;
    Node* bufA = scratch;
    Node* bufB = nullptr;
    for (int pass = 0;; ++pass) {
        const ReduceStep st = next_step(n, left, nslices);
        SliceGeom g;
        g.n_full = n; g.n_last = nl; g.in_stride = in_stride; g.nslices = nslices;
        if (st.kind == STEP_TAIL) {
            g.out_stride = 1;
            hipLaunchKernelGGL(reduce_tail_kernel, dim3(1, nslices), dim3(64), 0, stream, in, g, left, roots);
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
            hipLaunchKernelGGL(reduce_pass_kernel, dim3((uint32_t)grid, nslices), dim3(VKMR_PASS_WAVES * 64), 0, stream, in, g, out, m);
        } else {
            hipLaunchKernelGGL(reduce_collapse_kernel, dim3((uint32_t)ceil_shift(n, 7), nslices), dim3(64), 0, stream, in, g,
                               st.levels, out);
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
    if (count > 128 && !scratch_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_async: null scratch");
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
    if (count > 128 && !scratch_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_proof_async: null scratch");
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

size_t vkmr_hip_reduce_slices_scratch_bytes(uint64_t capacity, uint32_t nslices)
{
    if (nslices == 0) nslices = 1;
    return (size_t)scratch_cells(capacity, nslices) * nslices * sizeof(vkmr_digest);
}

vkmr_status vkmr_hip_reduce_slices_async(int dev, vkmr_stream s, const vkmr_digest* digests_dev, uint32_t nslices,
                                         uint64_t capacity, uint64_t count_last, uint32_t height, void* scratch_dev,
                                         vkmr_digest* roots_dev)
{
    if (!digests_dev || !roots_dev || nslices == 0) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_slices_async: bad argument");
    if (count_last == 0 || count_last > capacity || nslices > 65535u)
        return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_slices_async: bad slice geometry");
    if (!height_ok(capacity, height) || (nslices == 1 && !height_ok(count_last, height)))
        return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_slices_async: height does not reduce a slice to one node");
    if (capacity > 128 && !scratch_dev) return fail(VKMR_ERR_INVALID, "vkmr_hip_reduce_slices_async: null scratch");
    VKMR_TRY(hipSetDevice(dev));
    return reduce_launch(S(s), reinterpret_cast<const Node*>(digests_dev), nslices, nslices == 1 ? count_last : capacity,
                         count_last, height, reinterpret_cast<Node*>(scratch_dev), reinterpret_cast<Node*>(roots_dev));
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
        hipLaunchKernelGGL(reduce_level_kernel, dim3((uint32_t)grid), dim3(256), 0, S(s), in, n, out);
        VKMR_TRY(hipGetLastError());
        in = out;
        n = pairs;
    }
    if (height == 0) VKMR_TRY(hipMemcpyAsync(root_dev, digests_dev, sizeof(vkmr_digest), hipMemcpyDeviceToDevice, S(s)));
    return VKMR_OK;
}

// ---- combine --------------------------------------------------------------------

vkmr_status vkmr_hip_combine(int dev, const vkmr_digest* roots_host, uint32_t n, vkmr_digest* out_host)
{
    if (!roots_host || !out_host || n == 0) return fail(VKMR_ERR_INVALID, "vkmr_hip_combine: bad argument");
    VKMR_TRY(hipSetDevice(dev));
    uint32_t height = 1;   // at least one level: CpuSha256D::Root's do-while (SHA-256plus.cpp:515-547)
    while (ceil_shift(n, height) > 1) ++height;
    const size_t scratch = vkmr_hip_reduce_scratch_bytes(n);
    char* buf = nullptr;
    const size_t in_bytes = (size_t)n * sizeof(vkmr_digest);
    VKMR_TRY(hipMalloc(reinterpret_cast<void**>(&buf), in_bytes + scratch + sizeof(vkmr_digest)));
    vkmr_digest* d_in = reinterpret_cast<vkmr_digest*>(buf);
    void* d_scratch = buf + in_bytes;
    vkmr_digest* d_root = reinterpret_cast<vkmr_digest*>(buf + in_bytes + scratch);
    vkmr_status st = from_hip(hipMemcpy(d_in, roots_host, in_bytes, hipMemcpyHostToDevice), "hipMemcpy(roots)");
    if (st == VKMR_OK) st = vkmr_hip_reduce_async(dev, nullptr, d_in, n, height, d_scratch, d_root);
    if (st == VKMR_OK) st = from_hip(hipMemcpy(out_host, d_root, sizeof(vkmr_digest), hipMemcpyDeviceToHost), "hipMemcpy(root)");
    (void)hipFree(buf);
    return st;
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
