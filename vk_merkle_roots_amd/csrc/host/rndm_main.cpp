// rndm_main.cpp -- `rndm [seed] [number of strings] [max string length]`: same
// command line, same stream on stdout and same stderr notes as the reference tool
// (src/rndm/Rndm.cpp:20-71), produced from the restated generator with buffered
// output instead of one fwrite + fflush per byte/line.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <vector>

#include "rndm_stream.hpp"

int main(int argc, const char* argv[])
{
    const long seed = (argc > 1) ? std::atol(argv[1]) : (long)std::time(nullptr);
    std::cerr << "Using seed: " << seed << std::endl;
    if (argc < 3) {
        std::cerr << "Usage: rndm [seed] [number of strings] [max string length]" << std::endl;
        return 1;
    }
    const long bound = std::atol(argv[2]);
    const long max = (argc > 3) ? std::atol(argv[3]) : std::min(16384L, bound);
    if (max < 2) return 1;

    vkmr::GlibcRand gen((uint32_t)seed);
    std::vector<char> out;
    out.reserve(1 << 20);
    long count = 0, sum = 0;
    for (; count < bound; ++count) {
        const long len = 1 + (long)(gen.Next() % (uint32_t)(max - 1));
        for (long i = 0; i < len; ++i) out.push_back((char)(32 + gen.Next() % 94u));
        out.push_back('\n');
        sum += len;
        if (out.size() >= (1u << 20) - 20000u) {
            if (fwrite(out.data(), 1, out.size(), stdout) < out.size()) return 1;
            out.clear();
        }
    }
    if (!out.empty() && fwrite(out.data(), 1, out.size(), stdout) < out.size()) return 1;
    fflush(stdout);
    std::cerr << "Wrote " << count << " string(s) in a total of " << sum << " byte(s)." << std::endl << std::endl;
    return 0;
}
