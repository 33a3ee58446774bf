// pack_bench.cpp -- the host packer alone: a file of newline-separated strings, mapped, taken in spans, every span cut
// into parts that T threads index (pass 1) and pack (pass 2) into one buffer -- what Batch::PushLinesParallel does,
// without batches, copies or a GPU.  Prints GB/s and ns per line for both passes and for the one-pass-per-line forms
// they replaced (CountLines / PackLines).
//   g++ -O2 -std=c++17 -pthread -I vk_merkle_roots_amd/csrc/host -I include -o tools/pack_bench tools/pack_bench.cpp vk_merkle_roots_amd/csrc/host/stream_pack.cpp
//   tools/pack_bench file [threads=16] [span MiB=32] [resident=0]     resident=1: the file is read into memory first (no page faults
//                                                                      of the mapping inside the timed passes); 2: mapped, every span
//                                                                      populated with madvise(MADV_POPULATE_READ) before it is indexed;
//                                                                      3: the same on a helper thread, one span ahead
//   tools/pack_bench file threads span resident 1                    pass 2 with streaming stores
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "fork_join.hpp"
#include "stream_pack.hpp"

using namespace vkmr;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    const unsigned T = argc > 2 ? (unsigned)atoi(argv[2]) : 16;
    const size_t span = (size_t)(argc > 3 ? atol(argv[3]) : 32) << 20;
    const int fd = open(argv[1], O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) return 1;
    const size_t len = (size_t)st.st_size;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    const bool resident = mode == 1;
    const bool streaming = argc > 5 && atoi(argv[5]) != 0;
    std::vector<uint8_t> copy;
    if (resident) {
        copy.resize(len);
        size_t got = 0;
        while (got < len) {
            const ssize_t k = read(fd, copy.data() + got, len - got);
            if (k <= 0) return 1;
            got += (size_t)k;
        }
    }
    ForkJoin pool(T - 1);
    std::vector<uint32_t> data(span / 2 + (1 << 20));
    std::vector<vkmr_metadata> meta(span / 2 + 16);
    memset(data.data(), 1, data.size() * 4);
    memset(meta.data(), 1, meta.size() * sizeof(vkmr_metadata));
    for (int form = 0; form < 2; ++form) {
        for (int rep = 0; rep < 3; ++rep) {
            const uint8_t* b = resident ? copy.data() : static_cast<const uint8_t*>(mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0));
            if (b == MAP_FAILED) return 1;
            if (!resident) madvise(const_cast<uint8_t*>(b), len, MADV_SEQUENTIAL);
            std::vector<LineIndex> index(T);
            double t1 = 0, t2 = 0, t3 = 0;
            size_t lines = 0;
            const double t0 = now();
            std::thread ahead;
            for (size_t at = 0; at < len;) {
                size_t usable = len - at < span ? len - at : span;
#ifdef MADV_POPULATE_READ
                if (mode == 2) {
                    const double p0 = now();
                    const size_t lo_pg = at & ~(size_t)4095;
                    madvise(const_cast<uint8_t*>(b) + lo_pg, at + usable - lo_pg, MADV_POPULATE_READ);
                    t3 += now() - p0;
                }
                if (mode == 3) {
                    const double p0 = now();
                    if (ahead.joinable()) ahead.join();
                    else { const size_t lo_pg = at & ~(size_t)4095; madvise(const_cast<uint8_t*>(b) + lo_pg, at + usable - lo_pg, MADV_POPULATE_READ); }
                    t3 += now() - p0;
                    const size_t nxt = (at + usable) & ~(size_t)4095;
                    if (nxt < len) {
                        const size_t n2 = len - nxt < span + 8192 ? len - nxt : span + 8192;
                        ahead = std::thread([b, nxt, n2] { madvise(const_cast<uint8_t*>(b) + nxt, n2, MADV_POPULATE_READ); });
                    }
                }
#endif
                if (at + usable < len) {
                    const void* nl = memrchr(b + at, '\n', usable);
                    if (nl) usable = (size_t)(static_cast<const uint8_t*>(nl) - (b + at)) + 1;
                }
                struct Part { size_t lo, hi; LineCount c; };
                std::vector<Part> parts;
                size_t lo = 0;
                for (unsigned t = 0; t < T && lo < usable; ++t) {
                    size_t hi = (t + 1 == T) ? usable : usable / T * (t + 1);
                    if (hi < lo) hi = lo;
                    if (hi < usable) {
                        const void* nl = memchr(b + at + hi, '\n', usable - hi);
                        hi = nl ? (size_t)(static_cast<const uint8_t*>(nl) - (b + at)) + 1 : usable;
                    }
                    if (hi > lo) parts.push_back({lo, hi, {0, 0, 0, 0, false}});
                    lo = hi;
                }
                const double a = now();
                if (form == 0) pool.Run((unsigned)parts.size(), [&](unsigned t) { parts[t].c = IndexLines(b + at + parts[t].lo, parts[t].hi - parts[t].lo, &index[t]); });
                else pool.Run((unsigned)parts.size(), [&](unsigned t) { parts[t].c = CountLines(b + at + parts[t].lo, parts[t].hi - parts[t].lo); });
                const double c = now();
                std::vector<size_t> w0(parts.size()), c0(parts.size());
                size_t w = 0, n = 0;
                for (size_t t = 0; t < parts.size(); ++t) { w0[t] = w; c0[t] = n; w += parts[t].c.words; n += parts[t].c.strings; }
                if (w > data.size() || n > meta.size()) { printf("span does not fit\n"); return 1; }
                if (form == 0) pool.Run((unsigned)parts.size(), [&](unsigned t) { PackIndexed(b + at + parts[t].lo, parts[t].hi - parts[t].lo, index[t], data.data(), w0[t], w0[t] + parts[t].c.words, meta.data() + c0[t], nullptr, streaming); });
                else pool.Run((unsigned)parts.size(), [&](unsigned t) { PackLines(b + at + parts[t].lo, parts[t].hi - parts[t].lo, true, data.data(), w0[t], data.size(), meta.data() + c0[t], parts[t].c.strings); });
                const double d = now();
                t1 += c - a;
                t2 += d - c;
                lines += n;
                at += usable;
            }
            if (ahead.joinable()) ahead.join();
            const double total = now() - t0;
            printf("%s, %s, %u threads, spans of %zu MiB: %.1f ms = %.1f GB/s (%zu lines; pass 1 %.1f ms, pass 2 %.1f ms; %.2f + %.2f ns per line and thread)%s\n",
                   resident ? "resident" : mode == 2 ? "mapped+populate" : mode == 3 ? "mapped+populate ahead" : "mapped", form == 0 ? (streaming ? "index + pack-indexed (streaming stores)" : "index + pack-indexed") : "count + pack (per-line walk)", T, span >> 20, total * 1e3, (double)len / total / 1e9, lines, t1 * 1e3, t2 * 1e3,
                   t1 / (double)lines * 1e9 * T, t2 / (double)lines * 1e9 * T, mode >= 2 ? (" populate wait " + std::to_string(t3 * 1e3) + " ms").c_str() : "");
            if (!resident) munmap(const_cast<uint8_t*>(b), len);
        }
    }
    return 0;
}
