// ref_driver.cpp -- stdin driver for the UNMODIFIED reference CPU-serial backend.
//
// TEST INFRASTRUCTURE ONLY (see oracle/sha256d_oracle.c header).  The reference's
// own main (src/vkmr/Vkmr.cpp) cannot be compiled here because it includes the
// Vulkan headers (Vkmr.cpp:21 -> SHA-256vk.h -> <vulkan/vulkan.h>), so this file
// restates its run() loop (src/vkmr/Vkmr.cpp:28-58) over the reference's
// vkmr::Input and vkmr::CpuSha256D, which are compiled from the sources where
// they lie under /root/reference by oracle/Makefile.  Output line format is that
// of Vkmr.cpp:55 and :86.
#include <cstdio>
#include <iostream>
#include <string>

#include "Inputs.h"
#include "StopWatch.h"
#include "SHA-256plus.h"

int main()
{
    vkmr::CpuSha256D backend;
    std::cout << "Initializing for: " << backend.Name() << std::endl;

    vkmr::Input lines(stdin);
    StopWatch watch;
    watch.Start();
    size_t items = 0, octets = 0;
    for (; lines.Has();) {
        const std::string line = lines.Get();
        if (line.empty()) {
            std::cerr << "Read an empty string?" << std::endl;
            continue;
        }
        if (!backend.Add(line)) break;
        octets += line.size();
        items += 1;
    }
    if (items != 0) {
        const std::string root = backend.Root();
        const double ms = watch.Elapsed();
        std::cout << backend.Name() << ": computed root (of " << items << " item(s), " << octets
                  << " byte(s)) => " << root << " in " << ms << std::endl;
    }
    return 0;
}
