// reduce_plan.hpp -- the launch schedule of a slice reduction and its scratch budget.
// Host-only arithmetic (no HIP types), shared by the C ABI (vkmr_hip.hip) and by the
// CPU-side sweep in tests/c/reduce_plan_test.cpp, which checks that no schedule ever
// writes more scratch cells than vkmr_hip_reduce_scratch_bytes() promises.
//
// Takes the place of the pass loop of ReductionBySubgroup::GetCommandBuffer
// (reference src/vkmr/Reductions.cpp:472-521: `for (delta = 1; delta < applicable; delta *= 2S)`).
#pragma once
#include <stdint.h>

#ifndef VKMR_PASS_WAVES
#define VKMR_PASS_WAVES 4   // waves per workgroup in reduce_pass_kernel (2: 6 % slower, 8: the same -- profiles/r02_reduce_pass_shape.txt)
#endif
#ifndef VKMR_PASS_MAXM
#define VKMR_PASS_MAXM 3    // a wave consumes up to 2^3 chunks of 128 nodes: 4 levels per pass.  LDS for the pending halves is 2 KiB per level and wavefront:
                            // 3 levels = 6 wavefronts per SIMD, 4 = 5; with the issue-priority pass occupancy pays (4.60 vs 4.80 ms per 2^26, profiles/r03_ab_occupancy.txt)
#endif

#ifndef VKMR_COLLAPSE_WAVES
#define VKMR_COLLAPSE_WAVES 1   // wavefronts per workgroup of reduce_collapse_kernel (4, "one per SIMD of a CU": 121 vs 111 us, profiles/r03_reduce_top_kernels.txt)
#endif
#ifndef VKMR_TAIL_MAX
#define VKMR_TAIL_MAX 128   // reduce_tail_kernel takes up to this many nodes: 64 lanes, one wavefront (a 2048-node, 16-wavefront form
                            // was measured slower than collapsing down to 128 first: profiles/r03_reduce_top_kernels.txt)
#endif

namespace vkmr_plan {

inline uint64_t ceil_shift(uint64_t n, unsigned k) { return k >= 64 ? (n ? 1 : 0) : (n >> k) + ((n & ((1ull << k) - 1ull)) ? 1ull : 0ull); }

// Levels a bulk pass collapses for n input nodes per slice: the largest m+1 (m <= MAXM)
// that still leaves enough wavefronts (over all slices of the launch) to fill 256 CUs.
inline uint32_t pick_m(uint64_t n, uint32_t nslices)
{
    const uint64_t target_waves = 4096;
    for (int m = VKMR_PASS_MAXM; m > 0; --m)
        if (ceil_shift(n, 7 + m) * nslices >= target_waves && (128ull << m) <= n) return (uint32_t)m;
    return 0;
}

// One step of the reduction schedule for n nodes per slice with `left` levels to go:
//   bulk     n/128 >= 2048 wavefronts: reduce_pass_kernel, m+1 levels, every lane busy
//   tail     n <= TAIL_MAX (128): reduce_tail_kernel, one wavefront, all remaining levels
//   collapse otherwise: reduce_collapse_kernel, 7 levels, one wavefront per workgroup
struct Step { int kind; uint32_t levels; uint64_t n_out; };
enum { STEP_BULK = 0, STEP_COLLAPSE = 1, STEP_TAIL = 2 };

inline Step next_step(uint64_t n, uint32_t left, uint32_t nslices)
{
    Step st;
    if (n <= VKMR_TAIL_MAX) {
        st.kind = STEP_TAIL; st.levels = left; st.n_out = 1;
    } else if (ceil_shift(n, 7) * nslices >= 2048) {   // many slices keep every lane busy even on short runs
        st.kind = STEP_BULK; st.levels = pick_m(n, nslices) + 1u; st.n_out = ceil_shift(n, st.levels);
    } else {
        st.kind = STEP_COLLAPSE; st.levels = 7; st.n_out = ceil_shift(n, 7);
    }
    return st;
}

// Scratch cells per slice the schedule for exactly n nodes per slice writes: the outputs of
// its first two steps (ping-pong; every later step writes no more than the one two before it).
inline uint64_t cells_written(uint64_t n, uint32_t nslices)
{
    uint64_t total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const Step st = next_step(n, 64, nslices);
        if (st.kind == STEP_TAIL) break;   // the tail writes the root, not scratch
        n = st.n_out;
        total += n;
    }
    return total;
}

// Upper bound of cells_written(n, nslices) over EVERY n <= count.  Callers size scratch by
// `count` (a slice's capacity, a tree's leaf count) and then reduce shorter runs with it
// -- a short last slice, the sibling sub-trees of a proof -- whose own schedule can
// collapse fewer levels in its first pass than the full count's does.  First-pass output:
//   bulk with m = MAXM        ceil(n / 2^(MAXM+1))                       <= ceil(count / 32)
//   bulk with m <  MAXM       ceil(n / 2^(m+1)) with ceil(n / 2^(8+m)) * nslices < 4096,
//                             i.e. n <= (ceil(4096/nslices) - 1) * 2^(8+m)  => < 2^19 / nslices
//                             or with n < 2^(8+m) (too short for a longer walk)  => <= 128
//   collapse                  ceil(n / 128) with ceil(n/128) * nslices < 2048 => < 2^11 / nslices
// and never more than ceil(n / 2).  The second step at least halves the first one's output.
inline uint64_t cells_upper_bound(uint64_t count, uint32_t nslices)
{
    if (nslices == 0) nslices = 1;
    if (count <= VKMR_TAIL_MAX) return 2;
    const uint64_t half = ceil_shift(count, 1);
    uint64_t small = ((1ull << 19) + nslices - 1) / nslices;   // the m < MAXM and collapse regimes
    if (small < 128) small = 128;
    if (small > half) small = half;
    const uint64_t big = ceil_shift(count, VKMR_PASS_MAXM + 1);
    const uint64_t first = small > big ? small : big;
    return first + ceil_shift(first, 1) + 2;
}

}  // namespace vkmr_plan
