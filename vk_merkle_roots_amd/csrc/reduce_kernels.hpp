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

// Merkle proofs written DURING the reduction (the reference's to-do: "for an indicated leaf, allocate a buffer to hold and write
// out the intermediate values during reduction", README.md:118-120).  For each of `k` leaves the lane that hashes the pair the
// leaf's path node belongs to at tree level L stores the OTHER node of that pair (the node itself where it has no right
// sibling: the duplicate-last rule) to sib[q * height + L].  No extra hash; per hash step a wavefront makes k scalar range
// tests ("is path node q's pair one of the 64 this wavefront is forming?"), and the kernels are instantiated without all of it
// for reductions that want no proof.  `level0`: tree level of the launch's input nodes.
#define VKMR_MAX_PROOFS 16
struct ProofArgs { uint32_t k, height; uint64_t index[VKMR_MAX_PROOFS]; Node* sib; };

// The wavefront has just formed pairs of level-L nodes: lane i holds pair j_first + i * j_step as (l, r), when `valid`.
template <bool PROOFS>
__device__ __forceinline__ void note_siblings(const ProofArgs& pa, uint32_t L, uint64_t j_first, uint32_t j_step_log2, uint32_t lane, bool valid,
                                              const uint32_t (&l)[8], const uint32_t (&r)[8])
{
    if (!PROOFS) return;
    if (L >= pa.height) return;
    for (uint32_t q = 0; q < pa.k; ++q) {
        const uint64_t p = pa.index[q] >> L;                       // the path node at this level
        const uint64_t d = (p >> 1) - j_first;                     // its pair, counted from lane 0's (wraps to huge when below)
        if ((d >> j_step_log2) < 64u && (d & ((1ull << j_step_log2) - 1ull)) == 0ull) {   // wave-uniform
            if (lane == (uint32_t)(d >> j_step_log2) && valid) {
                Node t;
#pragma unroll
                for (int i = 0; i < 8; ++i) t.w[i] = (p & 1ull) ? l[i] : r[i];
                pa.sib[(size_t)q * pa.height + L] = t;
            }
        }
    }
}

// Does any proof's path run through the `span` level-`level0` nodes that begin at node `first`?  Asked ONCE per wavefront, so that
// the thousands of wavefronts no path touches skip every test below (k scalar loads per hash step otherwise: +2.4 % on a slice
// of 2^26 with 8 proofs, profiles/r04_proofs_in_the_pass.txt).
template <bool PROOFS>
__device__ __forceinline__ bool any_path_through(const ProofArgs& pa, uint32_t level0, uint64_t first, uint64_t span)
{
    if (!PROOFS) return false;
    bool mine = false;
    for (uint32_t q = 0; q < pa.k; ++q) mine = mine || ((pa.index[q] >> level0) - first) < span;
    return mine && blockIdx.y == 0;
}

// The same for the shuffle steps of the collapse / tail kernels, where the hashing lanes are 2, 4, 8 ... apart: every hashing lane
// tests its own pair (me, me + 1) of level-L nodes, x and r.
template <bool PROOFS>
__device__ __forceinline__ void note_sibling_of_lane(const ProofArgs& pa, uint32_t L, uint64_t me, bool hashing, const uint32_t (&x)[8], const uint32_t (&r)[8])
{
    if (!PROOFS) return;
    if (L >= pa.height) return;
    for (uint32_t q = 0; q < pa.k; ++q) {
        const uint64_t p = pa.index[q] >> L;
        if (hashing && (p >> 1) == (me >> 1)) {
            Node t;
#pragma unroll
            for (int i = 0; i < 8; ++i) t.w[i] = (p & 1ull) ? x[i] : r[i];
            pa.sib[(size_t)q * pa.height + L] = t;
        }
    }
}

template <bool PROOFS>
__device__ __forceinline__ void reduce_pass_body(const Node* __restrict__ in0, const SliceGeom& geom, Node* __restrict__ out0, uint32_t m,
                                                 const ProofArgs& pa, uint32_t level0)
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
    const bool mine = any_path_through<PROOFS>(pa, level0, base0, 128ull << m);
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
                if (PROOFS && mine) note_siblings<PROOFS>(pa, level0 + k, (first >> (k + 1)), 0u, lane, 2 * j < ck, l, r);
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

__global__ __launch_bounds__(VKMR_PASS_WAVES * 64) void reduce_pass_kernel(const Node* __restrict__ in0, SliceGeom geom,
                                                                           Node* __restrict__ out0, uint32_t m)
{
    ProofArgs none;
    none.k = 0u;
    reduce_pass_body<false>(in0, geom, out0, m, none, 0u);
}

__global__ __launch_bounds__(VKMR_PASS_WAVES * 64) void reduce_pass_proofs_kernel(const Node* __restrict__ in0, SliceGeom geom,
                                                                                  Node* __restrict__ out0, uint32_t m, ProofArgs pa, uint32_t level0)
{
    reduce_pass_body<true>(in0, geom, out0, m, pa, level0);
}

// Top of the tree: the last <= 128 nodes, exactly `levels` levels, ONE wavefront.  Level 1 comes from a coalesced pair
// load; the next six levels stay inside the wavefront with __shfl_down, exactly the shape of the reference's
// subgroupShuffleDown loop (src/shaders/SHA-256.comp:346-377).  Any levels left once a single node remains hash that
// node with itself ("keep iterating", README.md:94).
// (Round 3 tried a 16-wavefront form for up to 2048 nodes, to save a launch: its wavefronts share one CU's four SIMDs
// and each level costs 13.7 us against 9.4 us when reduce_collapse_kernel spreads them over the chip --
// profiles/r03_reduce_top_kernels.txt -- so the schedule collapses down to 128 nodes again and this stays one wavefront.)
template <bool PROOFS>
__device__ __forceinline__ void shuffle_collapse(uint32_t (&X)[8], uint64_t idx0, uint32_t lane, uint64_t n_in,
                                                 uint32_t& done, uint32_t levels, uint32_t steps, const ProofArgs& pa, uint32_t level0, bool mine)
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
        if (PROOFS && mine) note_sibling_of_lane<PROOFS>(pa, level0 + done, me, (lane & ((2u << t) - 1u)) == 0u && me < cnt, X, r);
        if ((lane & ((2u << t) - 1u)) == 0u && me < cnt) {
            uint32_t o[8];
            vkmr_dev::hash_pair(X, r, o);
#pragma unroll
            for (int i = 0; i < 8; ++i) X[i] = o[i];
        }
        ++done;
    }
}

template <bool PROOFS>
__device__ __forceinline__ void reduce_tail_body(const Node* __restrict__ in0, const SliceGeom& geom, uint32_t levels, Node* __restrict__ root0,
                                                 const ProofArgs& pa, uint32_t level0)
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
        if (PROOFS && blockIdx.y == 0) note_siblings<PROOFS>(pa, level0, 0ull, 0u, lane, true, a.w, b.w);
        vkmr_dev::hash_pair(a.w, b.w, X);
    }
    uint32_t done = 1;
    shuffle_collapse<PROOFS>(X, lane, lane, n_in, done, levels, 6, pa, level0, PROOFS && blockIdx.y == 0);
    while (done < levels) {   // a single node left: pair it with itself (wave-uniform loop)
        note_sibling_of_lane<PROOFS>(pa, level0 + done, 0ull, lane == 0 && blockIdx.y == 0, X, X);
        uint32_t o[8];
        vkmr_dev::hash_pair(X, X, o);
#pragma unroll
        for (int i = 0; i < 8; ++i) X[i] = o[i];
        ++done;
    }
    if (lane == 0) vkmr_dev::store_node(root, X);
}

__global__ __launch_bounds__(64) void reduce_tail_kernel(const Node* __restrict__ in0, SliceGeom geom, uint32_t levels,
                                                         Node* __restrict__ root0)
{
    ProofArgs none;
    none.k = 0u;
    reduce_tail_body<false>(in0, geom, levels, root0, none, 0u);
}

__global__ __launch_bounds__(64) void reduce_tail_proofs_kernel(const Node* __restrict__ in0, SliceGeom geom, uint32_t levels,
                                                                Node* __restrict__ root0, ProofArgs pa, uint32_t level0)
{
    reduce_tail_body<true>(in0, geom, levels, root0, pa, level0);
}

// Middle of the tree, where there are too few nodes to keep every SIMD busy: one
// wavefront per workgroup (so the wavefronts spread over all CUs) collapses 128
// nodes through `levels` (1..7) levels -- a coalesced pair load, then __shfl_down
// steps as in the reference's subgroup shader.  Lane utilisation is poor by
// construction here but these passes are latency-bound: what counts
// is the ~9 us one wavefront needs per level, not the idle lanes.
template <bool PROOFS>
__device__ __forceinline__ void reduce_collapse_body(const Node* __restrict__ in0, const SliceGeom& geom, uint32_t levels, Node* __restrict__ out0,
                                                     const ProofArgs& pa, uint32_t level0)
{
    const uint64_t n_in = slice_count(geom);
    const Node* __restrict__ in = in0 + blockIdx.y * geom.in_stride;
    Node* __restrict__ out = out0 + blockIdx.y * geom.out_stride;
    const uint32_t lane = threadIdx.x & 63u;
    // each wavefront of the workgroup collapses its own 128 nodes
    const uint64_t j0 = ((uint64_t)blockIdx.x * VKMR_COLLAPSE_WAVES + (threadIdx.x >> 6)) * 64u;
    const uint64_t j = j0 + lane;   // level-1 node of this lane
    const bool mine = any_path_through<PROOFS>(pa, level0, 2ull * j0, 128ull);
    uint32_t X[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (2 * j < n_in) {
        const Node a = vkmr_dev::load_node(in + 2 * j);
        const Node b = vkmr_dev::load_node(in + ((2 * j + 1 < n_in) ? 2 * j + 1 : 2 * j));
        if (PROOFS && mine) note_siblings<PROOFS>(pa, level0, j0, 0u, lane, true, a.w, b.w);
        vkmr_dev::hash_pair(a.w, b.w, X);
    }
    uint32_t done = 1;
    shuffle_collapse<PROOFS>(X, j, lane, n_in, done, levels, 6, pa, level0, mine);
    const uint64_t jo = j >> (levels - 1u);
    if ((lane & ((1u << (levels - 1u)) - 1u)) == 0u && jo < level_count(n_in, levels)) vkmr_dev::store_node(out + jo, X);
}

__global__ __launch_bounds__(VKMR_COLLAPSE_WAVES * 64) void reduce_collapse_kernel(const Node* __restrict__ in0, SliceGeom geom, uint32_t levels,
                                                                                     Node* __restrict__ out0)
{
    ProofArgs none;
    none.k = 0u;
    reduce_collapse_body<false>(in0, geom, levels, out0, none, 0u);
}

__global__ __launch_bounds__(VKMR_COLLAPSE_WAVES * 64) void reduce_collapse_proofs_kernel(const Node* __restrict__ in0, SliceGeom geom, uint32_t levels,
                                                                                            Node* __restrict__ out0, ProofArgs pa, uint32_t level0)
{
    reduce_collapse_body<true>(in0, geom, levels, out0, pa, level0);
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

