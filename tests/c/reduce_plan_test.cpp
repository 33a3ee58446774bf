// reduce_plan_test.cpp -- CPU-side sweep over the reduction schedule (csrc/reduce_plan.hpp):
// replays, for many (count, nslices), every launch vkmr_hip.hip would issue for a reduction,
// for a short last slice and for the sibling sub-trees of a proof, and checks that the scratch
// cells written never exceed what the size functions promise.  Built and run by
// tests/test_reduce_plan.py (no GPU).
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#include "reduce_plan.hpp"

using namespace vkmr_plan;

// Cells (per slice) a whole reduction of n nodes per slice touches, replaying the ping-pong of
// reduce_launch: bufA = [0, out1), bufB = [out1, out1 + out2), then A, B, ... again.
static uint64_t replay(uint64_t n, uint32_t nslices, int* steps)
{
    uint64_t high = 0, out1 = 0;
    uint32_t left = 63;
    for (int pass = 0;; ++pass) {
        const Step st = next_step(n, left, nslices);
        if (steps) ++*steps;
        if (st.kind == STEP_TAIL) return high;
        if (st.levels == 0 || st.n_out >= n) { printf("schedule does not shrink: n=%llu\n", (unsigned long long)n); exit(1); }
        uint64_t end;
        if (pass == 0) { out1 = st.n_out; end = out1; }
        else if (pass & 1) end = out1 + st.n_out;     // bufB
        else end = st.n_out;                           // bufA again
        if (pass >= 2 && (pass & 1) == 0 && st.n_out > out1) { printf("pass %d overruns bufA: n=%llu\n", pass, (unsigned long long)n); exit(1); }
        if (end > high) high = end;
        n = st.n_out;
        left -= st.levels < left ? st.levels : left;
    }
}

static unsigned long long g_checked = 0;

static void check(uint64_t n, uint32_t nslices, uint64_t budget_count)
{
    const uint64_t used = replay(n, nslices, nullptr);
    const uint64_t have = cells_upper_bound(budget_count, nslices);
    ++g_checked;
    if (used > have) {
        printf("OVERRUN: n=%llu nslices=%u writes %llu cells per slice, budget for count %llu is %llu\n", (unsigned long long)n, nslices,
               (unsigned long long)used, (unsigned long long)budget_count, (unsigned long long)have);
        exit(1);
    }
    if (used != 0 && cells_written(n, nslices) < used) {
        printf("cells_written underestimates: n=%llu nslices=%u\n", (unsigned long long)n, nslices);
        exit(1);
    }
}

int main()
{
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };

    // 1. every run of n <= count nodes against the budget for count, single slice: n near every
    //    regime boundary and random; this is what a short last slice and a proof's sub-trees do
    const uint64_t counts[] = {1, 2, 127, 128, 129, 1000, 1u << 15, (1u << 18) - 1, 1u << 18, (1u << 18) + 1, (1u << 20) - 256, 1u << 20,
                               3 * (1u << 20) - 256, (1u << 21) + 5, 1u << 22, (1u << 23) - 1, 1u << 23, (1u << 24) + 12345, 1u << 26,
                               (1ull << 29) + 7, 1ull << 33};
    for (uint64_t count : counts) {
        for (int k = 0; k <= 40; ++k) {
            const uint64_t p = 1ull << k;
            for (long d = -300; d <= 300; d += (d > -3 && d < 3) ? 1 : 99) {
                const uint64_t n = p + (uint64_t)d;
                if (n >= 1 && n <= count) check(n, 1, count);
            }
        }
        for (int i = 0; i < 20000; ++i) check(1 + rnd() % count, 1, count);
        check(count, 1, count);
    }

    // 2. the proof's own runs: for random (count, index) the sub-tree sizes min(2^l, count - lo)
    for (int i = 0; i < 3000; ++i) {
        const uint64_t count = (1ull << 18) + rnd() % ((1ull << 24) - (1ull << 18));
        const uint64_t index = rnd() % count;
        for (unsigned l = 1; ceil_shift(count, l - 1) > 1; ++l) {
            const uint64_t cl = ceil_shift(count, l), pnode = index >> l;
            uint64_t q = pnode ^ 1ull;
            if (q >= cl) q = pnode;
            const uint64_t lo = q << l;
            const uint64_t n = (count - lo < (1ull << l)) ? count - lo : (1ull << l);
            check(n, 1, count);
        }
    }

    // 3. several slices per launch: the schedule is the full slice's (capacity), the budget per slice
    //    is for (capacity, nslices); and the one-slice case runs the LAST slice's own count
    for (int lg = 1; lg <= 30; ++lg) {
        const uint64_t cap = 1ull << lg;
        const uint32_t ns_list[] = {1, 2, 3, 7, 8, 64, 1000, 4096, 4097, 32768};
        for (uint32_t ns : ns_list) {
            check(cap, ns, cap);
            if (ns == 1)
                for (int i = 0; i < 2000; ++i) check(1 + rnd() % cap, 1, cap);
        }
    }
    printf("ok: %llu schedules within budget\n", g_checked);
    return 0;
}
