// fork_join_test.cpp -- ThreadSanitizer check of the packer's fork-join pool and of the two-pass
// parallel packing scheme (IndexLines / PackIndexed, as Batch::PushLinesParallel runs them) on plain memory (built and run by tests/test_host_fuzz.py).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "fork_join.hpp"
#include "stream_pack.hpp"

using namespace vkmr;

int main()
{
    ForkJoin pool(5);
    // 1. many small forks: every task runs exactly once per Run
    std::vector<unsigned> hits(pool.Width(), 0);
    for (int round = 0; round < 2000; ++round) {
        const unsigned n = 1 + round % pool.Width();
        pool.Run(n, [&](unsigned i) { hits[i] += 1; });
    }
    unsigned long total = 0;
    for (unsigned h : hits) total += h;
    unsigned long want = 0;
    for (int round = 0; round < 2000; ++round) want += 1 + round % pool.Width();
    if (total != want) { printf("task count %lu != %lu\n", total, want); return 1; }

    // 2. two-pass parallel packing equals sequential packing
    std::string text;
    for (int i = 0; i < 200000; ++i) { text.append((size_t)(i * 7 % 131), (char)('a' + i % 26)); text.push_back('\n'); }
    const uint8_t* b = reinterpret_cast<const uint8_t*>(text.data());
    const size_t len = text.size();
    std::vector<uint32_t> seq(len / 4 + 300000), par(len / 4 + 300000);
    std::vector<vkmr_metadata> mseq(200001), mpar(200001);
    const PackResult rs = PackLines(b, len, true, seq.data(), 0, seq.size(), mseq.data(), mseq.size());
    struct Part { size_t lo, hi; LineCount c; };
    std::vector<Part> parts;
    const unsigned T = pool.Width();
    size_t lo = 0;
    for (unsigned t = 0; t < T; ++t) {
        size_t hi = (t + 1 == T) ? len : len / T * (t + 1);
        if (hi < len) { const void* nl = memchr(b + hi, '\n', len - hi); hi = nl ? (size_t)((const uint8_t*)nl - b) + 1 : len; }
        parts.push_back({lo, hi, {0, 0, 0, 0, false}});
        lo = hi;
    }
    std::vector<LineIndex> index(T);
    pool.Run(T, [&](unsigned t) { parts[t].c = IndexLines(b + parts[t].lo, parts[t].hi - parts[t].lo, &index[t]); });
    std::vector<size_t> w0(T), c0(T);
    size_t w = 0, c = 0;
    for (unsigned t = 0; t < T; ++t) { w0[t] = w; c0[t] = c; w += parts[t].c.words; c += parts[t].c.strings; }
    // every part within its own words: the vector copies of one part must not reach into the next (another thread's)
    pool.Run(T, [&](unsigned t) { PackIndexed(b + parts[t].lo, parts[t].hi - parts[t].lo, index[t], par.data(), w0[t], w0[t] + parts[t].c.words, mpar.data() + c0[t]); });
    if (c != rs.strings || w != rs.words) { printf("counts differ\n"); return 1; }
    if (memcmp(seq.data(), par.data(), w * 4) != 0 || memcmp(mseq.data(), mpar.data(), c * sizeof(vkmr_metadata)) != 0) { printf("packed data differ\n"); return 1; }
    printf("ok %zu strings %zu words\n", c, w);
    return 0;
}
