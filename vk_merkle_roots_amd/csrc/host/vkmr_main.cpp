// vkmr_main.cpp -- `vkmr [backend]`: newline-separated strings on stdin, Merkle root of
// their SHA-256d hashes on stdout.
//
// Same command line and output as the reference's main/run (src/vkmr/Vkmr.cpp:28-97):
// backend "CPU" or a device name -- here "hip:<n>" / "hip:all" instead of a Vulkan
// device name; with no argument and more than one backend it lists them and exits 1.
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <iostream>
#include <string>

#include "cpu_sha256d.hpp"
#include "hip_sha256d.hpp"
#include "inputs.hpp"
#include "timing.hpp"
#include "util.hpp"

// The input loop (reference run(), src/vkmr/Vkmr.cpp:28-58).
static int run(vkmr::ISha256D& backend)
{
    vkmr::Input input(stdin);
    size_t size = 0, count = 0;
    vkmr::StopWatch sw;
    sw.Start();
    bool refused = false;
    while (input.Has() && !refused) {
        const char* p = nullptr;
        size_t n = 0;
        bool final = false;
        {
            vkmr::timing::Scope ts(vkmr::timing::READ);
            input.GetBlock(&p, &n, &final);
        }
        vkmr::ISha256D::Tally tally;
        refused = !backend.AddLines(p, n, final, &tally);
        // the reference reads one more, empty, string when the stream ends right after a '\n'
        // (or is empty): Input::Has is "not at EOF yet" (src/vkmr/Inputs.cpp:52-54)
        if (final && !refused && (n == 0 || p[n - 1] == '\n')) ++tally.empties;
        for (size_t i = 0; i < tally.empties; ++i) std::cerr << "Read an empty string?" << std::endl;
        size += tally.bytes;
        count += tally.items;
    }
    if (const int err = input.Error()) {
        // the stream did not end, it broke (EIO, a descriptor gone bad): a root over what arrived would look like an answer
        std::cerr << "Reading the input failed after " << count << " item(s): " << strerror(err) << "; no root." << std::endl;
        return 2;
    }
    if (count > 0) {
        const std::string root = backend.Root();
        const double elapsed = sw.Elapsed();
        std::cout << backend.Name() << ": computed root (of " << count << " item(s), " << size << " byte(s)) => " << root
                  << " in " << elapsed << std::endl;
        // the Merkle proof of one leaf, when asked for (VKMR_PROOF_INDEX; the reference's to-do, README.md:118-120)
        if (auto* hip = dynamic_cast<vkmr::HipSha256D::Instance*>(&backend))
            for (const auto& l : hip->ProofLines()) std::cout << l << std::endl;
        vkmr::timing::Report(std::cerr);
    }
    return 0;
}

// VKMR_TIMING=1: milliseconds since main() began at the points where a short run spends its time outside the stopwatch
// (device enumeration, backend construction, exit), on stderr.
static void stamp(const char* what)
{
    static const bool on = getenv("VKMR_TIMING") != nullptr;
    static const auto t0 = std::chrono::steady_clock::now();
    if (on) std::cerr << "[timing] " << what << ": " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() << " ms" << std::endl;
}

int main(int argc, const char* argv[])
{
    stamp("main");
    vkmr::CpuSha256D cpu;
    vkmr::HipSha256D gpus;
    std::string choice;
    if (argc > 1) {
        choice = argv[1];
    } else {
        std::vector<std::string> available = gpus.Available();
        available.insert(available.begin(), cpu.Name());
        if (available.size() == 1) {
            choice = available.front();
        } else {
            std::cerr << "Usage: " << argv[0] << " <name of compute device>" << std::endl;
            std::cerr << "Available: " << std::endl;
            for (const auto& name : available) std::cerr << "* " << name << gpus.Describe(name) << std::endl;
            return 1;
        }
    }
    std::cout << "Initializing for: " << choice << std::endl;
    if (cpu.Name() == choice) return run(cpu);   // before any question to the GPU runtime: `vkmr CPU` makes no HIP call
    if (gpus.Has(choice)) {
        stamp("devices enumerated");
        vkmr::HipConfig cfg = vkmr::HipConfig::FromEnv();
        struct stat st;
        if (fstat(STDIN_FILENO, &st) == 0 && S_ISREG(st.st_mode)) {   // `vkmr hip:all < file`: the size of the input is known
            const off_t at = lseek(STDIN_FILENO, 0, SEEK_CUR);
            if (st.st_size > (at > 0 ? at : 0)) cfg.expected_input_bytes = (uint64_t)(st.st_size - (at > 0 ? at : 0));
        }
        auto instance = gpus.Get(choice, cfg);
        stamp("backend constructed");
        const int rc = run(*instance);
        stamp("root printed");
#if !defined(__SANITIZE_ADDRESS__) && !defined(__SANITIZE_THREAD__) && !defined(VKMR_ORDERLY_EXIT)
        // The root is printed: what is left is giving pinned buffers, HBM, streams and events back one call at a time
        // (hipHostFree alone: 40 ms of a 150 ms run) just before the process ends and the driver reclaims all of it
        // at once.  Skip the destructors; VKMR_ORDERLY_EXIT=1 (or a sanitizer build) keeps them.
        // (a profiler writes its output from exit handlers: under rocprofv3 -- it preloads its tool library -- or with
        // VKMR_ORDERLY_EXIT set, leave the ordinary way)
        const char* preload = getenv("LD_PRELOAD");
        if (!getenv("VKMR_ORDERLY_EXIT") && !getenv("ROCP_TOOL_LIBRARIES") && !(preload && preload[0])) {
            std::cout.flush();
            std::cerr.flush();
            fflush(nullptr);
            _exit(rc);
        }
#endif
        return rc;
    }
    std::cerr << "No device selected; aborting." << std::endl;
    return 1;
}
