// rndm_stream.cpp -- see rndm_stream.hpp.
#include "rndm_stream.hpp"

#include <cstring>

#include "vkmr_hip.h"

namespace vkmr {

// glibc srandom_r for TYPE_3 (degree 31, separation 3): a Lehmer fill of the
// table, then 310 discarded draws.
GlibcRand::GlibcRand(uint32_t seed)
{
    if (seed == 0) seed = 1;
    int32_t word = (int32_t)seed;
    m_r[0] = (uint32_t)word;
    for (int i = 1; i < 31; ++i) {
        const long hi = word / 127773;
        const long lo = word % 127773;
        long next = 16807 * lo - 2836 * hi;
        if (next < 0) next += 2147483647;
        word = (int32_t)next;
        m_r[i] = (uint32_t)word;
    }
    m_front = 3;
    m_rear = 0;
    for (int i = 0; i < 310; ++i) (void)Next();
}

}  // namespace vkmr

extern "C" {

// First n values of rand() after srand(seed) (test hook).
__attribute__((visibility("default"))) void vkmr_host_rndm_rand(uint32_t seed, int32_t* out, uint64_t n)
{
    vkmr::GlibcRand g(seed);
    for (uint64_t i = 0; i < n; ++i) out[i] = (int32_t)g.Next();
}

// Stateful form: the stream of `rndm seed * maxlen` produced batch after batch (a packed batch
// addresses at most 2^32 words, so long-string workloads need several).
__attribute__((visibility("default"))) void* vkmr_host_rndm_open(uint32_t seed) { return new vkmr::GlibcRand(seed); }
__attribute__((visibility("default"))) void vkmr_host_rndm_close(void* h) { delete static_cast<vkmr::GlibcRand*>(h); }

// Next `count` strings of the stream into a fresh packed batch (metadata starts at word 0).
__attribute__((visibility("default"))) int64_t vkmr_host_rndm_next(void* h, uint64_t count, uint32_t maxlen, uint32_t* data,
                                                                    uint64_t data_capacity_words, vkmr_metadata* meta, uint64_t* words_used)
{
    if (!h || maxlen < 2 || !data || !meta) return -1;
    vkmr::GlibcRand& g = *static_cast<vkmr::GlibcRand*>(h);
    uint64_t w = 0;
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t len = 1u + g.Next() % (maxlen - 1u);
        const uint64_t nw = (len + 3u) / 4u;
        if (w + nw > data_capacity_words || w > 0xFFFFFFFFull) return -1;
        meta[i].start = (uint32_t)w;
        meta[i].size = len;
        uint8_t* dst = reinterpret_cast<uint8_t*>(data + w);
        data[w + nw - 1] = 0u;
        for (uint32_t b = 0; b < len; ++b) dst[b] = (uint8_t)(32u + g.Next() % 94u);
        w += nw;
    }
    if (words_used) *words_used = w;
    return (int64_t)count;
}

// Generates the strings of `rndm seed count maxlen` directly in the packed batch
// layout (Batch::Push, reference src/vkmr/Batches.cpp:64-121): string i starts at the
// word after string i-1, `start` is a word index, bytes past `size` in the last word
// are zero.  Returns the number of strings written (== count) or -1 when the data
// buffer is too small; *words_used receives the packed length in words.
__attribute__((visibility("default"))) int64_t vkmr_host_rndm_pack(uint32_t seed, uint64_t count, uint32_t maxlen,
                                                                    uint32_t* data, uint64_t data_capacity_words,
                                                                    vkmr_metadata* meta, uint64_t* words_used)
{
    if (maxlen < 2 || !data || !meta) return -1;
    vkmr::GlibcRand g(seed);
    uint64_t w = 0;
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t len = 1u + g.Next() % (maxlen - 1u);
        const uint64_t nw = (len + 3u) / 4u;
        if (w + nw > data_capacity_words || w > 0xFFFFFFFFull) return -1;
        meta[i].start = (uint32_t)w;
        meta[i].size = len;
        uint8_t* dst = reinterpret_cast<uint8_t*>(data + w);
        data[w + nw - 1] = 0u;
        for (uint32_t b = 0; b < len; ++b) dst[b] = (uint8_t)(32u + g.Next() % 94u);
        w += nw;
    }
    if (words_used) *words_used = w;
    return (int64_t)count;
}

}  // extern "C"
