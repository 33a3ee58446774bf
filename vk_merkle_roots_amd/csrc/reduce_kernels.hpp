// reduce_kernels.hpp -- the REDUCE kernels: pairwise SHA-256d tree over a slice of digests (gfx950).
// Replace the reference's shader entries `_SHA_256_2_BE_` with and without `_VKMR_BY_SUBGROUP_`
// (src/shaders/SHA-256.comp:308-434).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "reduce_plan.hpp"   // VKMR_PASS_WAVES, VKMR_PASS_MAXM and the host-side schedule
#include "sha256d_device.hpp"
#include "stamps.hpp"

using vkmr_dev::Node;

// ============================================================================
// REDUCE
// ============================================================================

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
    VKMR_STAMP(t_begin);
    VKMR_STAMP_RT(rt_begin);

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
                        // both nodes in one go: the four 16-byte loads issue back to back, so the 128-byte
                        // line two lanes share is requested once.  (A conditional second load is compiled as
                        // load - wait - branch - load; by then the line has left L1 and often L2:
                        // 1.5x the algorithmic HBM reads, profiles/r02_reduce_fetch.txt.)  An odd count
                        // pairs the last node with itself: it is simply read twice.
                        const uint64_t jb = (2 * j + 1 < ck) ? 2 * j + 1 : 2 * j;
                        const Node a = vkmr_dev::load_node(in + 2 * j);
                        const Node b = vkmr_dev::load_node(in + jb);
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
                    // the pending half was written by OTHER lanes of this wavefront (below, an
                    // earlier chunk): order those LDS writes before these reads explicitly
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
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
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    const uint64_t jo = (base0 >> (m + 1)) + lane;
    if (jo < level_count(n_in, m + 1)) vkmr_dev::store_node(out + jo, X);
#ifdef VKMR_STAMPS
    {
        unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < VKMR_STAMP_SLOTS) {
            unsigned long long* o = g_stamps + (size_t)blockIdx.x * 8;
            o[0] = t_begin; o[1] = rt_begin; o[2] = t_begin; o[3] = t_begin; o[4] = t_end; o[5] = rt_end;
            o[6] = 0x524544ull /* "RED" */; o[7] = gridDim.x;
        }
    }
#endif
}

// Top of the tree: the last <= 128 nodes, exactly `levels` levels, ONE wavefront.  Level 1 comes from a coalesced pair
// load; the next six levels stay inside the wavefront with __shfl_down, exactly the shape of the reference's
// subgroupShuffleDown loop (src/shaders/SHA-256.comp:346-377).  Any levels left once a single node remains hash that
// node with itself ("keep iterating", README.md:94).
// (Round 3 tried a 16-wavefront form for up to 2048 nodes, to save a launch: its wavefronts share one CU's four SIMDs
// and each level costs 13.7 us against 9.4 us when reduce_collapse_kernel spreads them over the chip --
// profiles/r03_reduce_top_kernels.txt -- so the schedule collapses down to 128 nodes again and this stays one wavefront.)
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

__global__ __launch_bounds__(64) void reduce_tail_kernel(const Node* __restrict__ in0, SliceGeom geom, uint32_t levels,
                                                         Node* __restrict__ root0)
{
    const uint32_t n_in = (uint32_t)slice_count(geom);
    const Node* __restrict__ in = in0 + blockIdx.y * geom.in_stride;
    Node* __restrict__ root = root0 + blockIdx.y * geom.out_stride;
    const uint32_t lane = threadIdx.x;
    uint32_t X[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (levels == 0) {   // n_in == 1: the root is the node itself
        if (lane == 0) *root = in[0];
        return;
    }
    if (2 * lane < n_in) {
        const Node a = vkmr_dev::load_node(in + 2 * lane);
        const Node b = vkmr_dev::load_node(in + ((2 * lane + 1 < n_in) ? 2 * lane + 1 : 2 * lane));
        vkmr_dev::hash_pair(a.w, b.w, X);
    }
    uint32_t done = 1;
    shuffle_collapse(X, lane, lane, n_in, done, levels, 6);
    while (done < levels) {   // a single node left: pair it with itself (wave-uniform loop)
        uint32_t o[8];
        vkmr_dev::hash_pair(X, X, o);
#pragma unroll
        for (int i = 0; i < 8; ++i) X[i] = o[i];
        ++done;
    }
    if (lane == 0) vkmr_dev::store_node(root, X);
}

// Middle of the tree, where there are too few nodes to keep every SIMD busy: one
// wavefront per workgroup (so the wavefronts spread over all CUs) collapses 128
// nodes through `levels` (1..7) levels -- a coalesced pair load, then __shfl_down
// steps as in the reference's subgroup shader.  Lane utilisation is poor by
// construction here but these passes are latency-bound: what counts
// is the ~9 us one wavefront needs per level, not the idle lanes.
__global__ __launch_bounds__(VKMR_COLLAPSE_WAVES * 64) void reduce_collapse_kernel(const Node* __restrict__ in0, SliceGeom geom, uint32_t levels,
                                                                                     Node* __restrict__ out0)
{
    const uint64_t n_in = slice_count(geom);
    const Node* __restrict__ in = in0 + blockIdx.y * geom.in_stride;
    Node* __restrict__ out = out0 + blockIdx.y * geom.out_stride;
    const uint32_t lane = threadIdx.x & 63u;
    // each wavefront of the workgroup collapses its own 128 nodes
    const uint64_t j = ((uint64_t)blockIdx.x * VKMR_COLLAPSE_WAVES + (threadIdx.x >> 6)) * 64u + lane;   // level-1 node of this lane
    uint32_t X[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (2 * j < n_in) {
        const Node a = vkmr_dev::load_node(in + 2 * j);
        const Node b = vkmr_dev::load_node(in + ((2 * j + 1 < n_in) ? 2 * j + 1 : 2 * j));
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
    const Node b = vkmr_dev::load_node(in + ((2 * p + 1 < n_in) ? 2 * p + 1 : 2 * p));
    uint32_t o[8];
    vkmr_dev::hash_pair(a.w, b.w, o);
    vkmr_dev::store_node(out + p, o);
}

