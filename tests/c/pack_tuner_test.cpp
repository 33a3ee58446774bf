// pack_tuner_test.cpp -- PackTuner (csrc/host/batches.hpp): which form of the packer's second pass the next call uses.
// Built and run by tests/test_host_tools.py.
#include <cstdio>

#include "batches.hpp"

using vkmr::PackTuner;

// Feeds the tuner calls of 32 MiB whose two forms take `ord` and `str` seconds; returns how many of `calls` used streaming stores.
static unsigned drive(PackTuner& t, unsigned calls, double ord, double str)
{
    unsigned streaming = 0;
    for (unsigned i = 0; i < calls; ++i) {
        const bool s = t.Next();
        streaming += s;
        t.Report(s, (size_t)32 << 20, s ? str : ord);
    }
    return streaming;
}

int main()
{
    {   // the first eight calls alternate; after that the faster form is used, the other one every sixteenth call
        PackTuner t;
        if (drive(t, 8, 1.0e-3, 0.5e-3) != 4) { printf("the first eight calls must alternate\n"); return 1; }
        const unsigned s = drive(t, 160, 1.0e-3, 0.5e-3);
        if (s != 150) { printf("streaming faster: %u of 160 calls streamed, expected 150\n", s); return 1; }
    }
    {
        PackTuner t;
        drive(t, 8, 0.5e-3, 1.0e-3);
        const unsigned s = drive(t, 160, 0.5e-3, 1.0e-3);
        if (s != 10) { printf("ordinary faster: %u of 160 calls streamed, expected 10\n", s); return 1; }
    }
    {   // the host's state changes: the probes notice, and the choice follows within a few of them
        PackTuner t;
        drive(t, 40, 1.0e-3, 0.5e-3);
        drive(t, 16 * 12, 0.5e-3, 2.0e-3);   // streaming is now four times slower
        const unsigned s = drive(t, 160, 0.5e-3, 2.0e-3);
        if (s != 10) { printf("after the change: %u of 160 calls streamed, expected 10\n", s); return 1; }
    }
    {   // forced either way; calls that moved nothing do not count
        PackTuner never(0), always(1);
        if (drive(never, 50, 1.0, 1.0e-6) != 0 || drive(always, 50, 1.0e-6, 1.0) != 50) { printf("forced modes\n"); return 1; }
        PackTuner t;
        t.Report(true, 0, 1.0);
        t.Report(true, 100, 0.0);
        if (t.Rate(true) != 0.0) { printf("empty reports must not count\n"); return 1; }
    }
    printf("ok\n");
    return 0;
}
