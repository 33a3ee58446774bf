// timing.hpp -- VKMR_TIMING=1: where the calling thread of the stream processor spends a run, phase by phase (stderr,
// after the result line).  The reference has one stopwatch around the whole run (src/vkmr/Vkmr.cpp:36-55); this breaks
// that figure down: reading, the two passes of the parallel packer, waits for a batch / a slice, launches, the drain.
#pragma once
#include <chrono>
#include <cstdlib>
#include <ostream>

namespace vkmr {
namespace timing {

enum Phase { READ, INDEX, PACK, PACK_SERIAL, BATCH, SLICE, MAP_WAIT, MAP, MAP_COPIES, MAP_LAUNCH, UPDATE, DRAIN_MAP, DRAIN_REDUCE, PHASES };

inline bool On()
{
    static const bool on = getenv("VKMR_TIMING") != nullptr;
    return on;
}

struct Table { double ms[PHASES] = {}, longest[PHASES] = {}, first[PHASES] = {}; unsigned long calls[PHASES] = {}, slow[PHASES] = {}; };
inline Table& Totals()
{
    static Table t;
    return t;
}

// Adds the life time of the object to a phase (the calling thread's only: the packer's workers are not timed, the
// fork-join they run in is).
class Scope {
public:
    explicit Scope(Phase p) : m_p(p), m_on(On())
    {
        if (m_on) m_t0 = std::chrono::steady_clock::now();
    }
    ~Scope()
    {
        if (!m_on) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - m_t0).count();
        Table& t = Totals();
        t.ms[m_p] += ms;
        if (ms > t.longest[m_p]) t.longest[m_p] = ms;
        if (ms > 0.5) ++t.slow[m_p];
        if (t.calls[m_p]++ == 0) t.first[m_p] = ms;
    }

private:
    Phase m_p;
    bool m_on;
    std::chrono::steady_clock::time_point m_t0;
};

inline void Report(std::ostream& os)
{
    if (!On()) return;
    static const char* names[PHASES] = {"read (next span of stdin)", "pack pass 1 (index the lines, fork-join)", "pack pass 2 (copy the lines, fork-join)",
                                        "pack, serial remainder", "wait for / allocate a batch", "wait for / allocate a slice",
                                        "wait for the oldest mapping (pipeline full)", "map dispatch (copies + launch)", "  of which the two copies", "  of which the launch", "poll mappings and reductions", "drain: last batch and mappings",
                                        "drain: reductions and root"};
    for (int p = 0; p < PHASES; ++p)
        os << "[timing] " << names[p] << ": " << Totals().ms[p] << " ms in " << Totals().calls[p] << " call(s); first " << Totals().first[p] << ", longest "
           << Totals().longest[p] << ", " << Totals().slow[p] << " over 0.5 ms" << std::endl;
}

}  // namespace timing
}  // namespace vkmr
