// stream_pack.cpp -- see stream_pack.hpp.
#include "stream_pack.hpp"

#include <cstdlib>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace vkmr {

// ---- portable forms: one memchr per line -------------------------------------------------------------------

PackResult PackLinesPortable(const uint8_t* buf, size_t len, bool final, uint32_t* data, uint64_t first_word,
                             uint64_t data_capacity_words, vkmr_metadata* meta, uint64_t meta_capacity)
{
    PackResult r = {0, 0, 0, 0, 0};
    uint64_t w = first_word;
    size_t pos = 0;
    while (pos < len) {
        const uint8_t* nl = static_cast<const uint8_t*>(memchr(buf + pos, '\n', len - pos));
        if (!nl && !final) break;   // incomplete line: wait for more input
        const size_t end = nl ? (size_t)(nl - buf) : len;
        const size_t n = end - pos;
        if (n == 0) {
            ++r.empties;
        } else {
            const uint64_t nw = (n + 3u) / 4u;
            if (r.strings == meta_capacity || w + nw > data_capacity_words || w > 0xFFFFFFFFull || n > 0xFFFFFFFFull)
                break;
            meta[r.strings].start = (uint32_t)w;
            meta[r.strings].size = (uint32_t)n;
            data[w + nw - 1] = 0u;
            memcpy(data + w, buf + pos, n);
            w += nw;
            ++r.strings;
            r.bytes += n;
        }
        pos = nl ? end + 1 : end;
    }
    r.consumed = pos;
    r.words = w - first_word;
    return r;
}

LineCount CountLinesPortable(const uint8_t* buf, size_t len)
{
    LineCount c = {0, 0, 0, 0, false};
    size_t pos = 0;
    while (pos < len) {
        const uint8_t* nl = static_cast<const uint8_t*>(memchr(buf + pos, '\n', len - pos));
        const size_t end = nl ? (size_t)(nl - buf) : len;
        const size_t n = end - pos;
        if (n == 0) {
            ++c.empties;
        } else {
            if (n > 0xFFFFFFFFull) c.too_long = true;
            ++c.strings;
            c.words += (n + 3u) / 4u;
            c.bytes += n;
        }
        pos = nl ? end + 1 : end;
    }
    return c;
}

namespace {

// ---- AVX2 forms: the newlines of 64 input bytes at a time -----------------------------------------------------
// Lines of rndm-like streams are ~64 bytes: a memchr call per line costs more than the line's copy.  Two 32-byte compares
// give a 64-bit mask of the newline positions of a block; the lines are then walked bit by bit.  Same results as the
// portable forms, line for line (tests/test_host_tools.py, test_host_fuzz.py compare them).

#if defined(__x86_64__)
#define VKMR_HAVE_AVX2_PATH 1

__attribute__((target("avx2"))) inline uint64_t newline_mask64(const uint8_t* p)
{
    const __m256i nl = _mm256_set1_epi8('\n');
    const uint32_t lo = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p)), nl));
    const uint32_t hi = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + 32)), nl));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

__attribute__((target("avx2"))) LineCount CountLinesAvx2(const uint8_t* buf, size_t len)
{
    LineCount c = {0, 0, 0, 0, false};
    size_t line = 0;   // start of the current line
    size_t blk = 0;
    for (; blk + 64 <= len; blk += 64) {
        uint64_t m = newline_mask64(buf + blk);
        while (m) {
            const size_t end = blk + (size_t)__builtin_ctzll(m);
            m &= m - 1;
            const size_t n = end - line;
            if (n == 0) {
                ++c.empties;
            } else {
                if (n > 0xFFFFFFFFull) c.too_long = true;
                ++c.strings;
                c.words += (n + 3u) / 4u;
                c.bytes += n;
            }
            line = end + 1;
        }
    }
    // the last, partial block -- and the line that began before it -- the portable way
    const size_t from = line;
    const LineCount t = CountLinesPortable(buf + from, len - from);
    c.strings += t.strings; c.words += t.words; c.bytes += t.bytes; c.empties += t.empties;
    c.too_long = c.too_long || t.too_long;
    return c;
}

__attribute__((target("avx2"))) PackResult PackLinesAvx2(const uint8_t* buf, size_t len, bool final, uint32_t* data, uint64_t first_word,
                                                         uint64_t data_capacity_words, vkmr_metadata* meta, uint64_t meta_capacity)
{
    PackResult r = {0, 0, 0, 0, 0};
    uint64_t w = first_word;
    size_t line = 0;
    bool full = false;
    size_t blk = 0;
    for (; blk + 64 <= len && !full; blk += 64) {
        uint64_t m = newline_mask64(buf + blk);
        while (m) {
            const size_t end = blk + (size_t)__builtin_ctzll(m);
            m &= m - 1;
            const size_t n = end - line;
            if (n == 0) {
                ++r.empties;
            } else {
                const uint64_t nw = (n + 3u) / 4u;
                if (r.strings == meta_capacity || w + nw > data_capacity_words || w > 0xFFFFFFFFull || n > 0xFFFFFFFFull) {
                    full = true;   // this line does not fit: it stays unconsumed
                    break;
                }
                meta[r.strings].start = (uint32_t)w;
                meta[r.strings].size = (uint32_t)n;
                data[w + nw - 1] = 0u;
                memcpy(data + w, buf + line, n);
                w += nw;
                ++r.strings;
                r.bytes += n;
            }
            line = end + 1;
        }
    }
    r.consumed = line;
    r.words = w - first_word;
    if (full) return r;
    // the last, partial block -- and the line that began before it -- the portable way
    const PackResult t = PackLinesPortable(buf + line, len - line, final, data, w, data_capacity_words, meta + r.strings, meta_capacity - r.strings);
    r.consumed += t.consumed; r.strings += t.strings; r.words += t.words; r.bytes += t.bytes; r.empties += t.empties;
    return r;
}

bool have_avx2()
{
    static const bool yes = __builtin_cpu_supports("avx2") && !getenv("VKMR_NO_AVX2");
    return yes;
}
#endif

}  // namespace

PackResult PackLines(const uint8_t* buf, size_t len, bool final, uint32_t* data, uint64_t first_word,
                     uint64_t data_capacity_words, vkmr_metadata* meta, uint64_t meta_capacity)
{
#ifdef VKMR_HAVE_AVX2_PATH
    if (have_avx2()) return PackLinesAvx2(buf, len, final, data, first_word, data_capacity_words, meta, meta_capacity);
#endif
    return PackLinesPortable(buf, len, final, data, first_word, data_capacity_words, meta, meta_capacity);
}

LineCount CountLines(const uint8_t* buf, size_t len)
{
#ifdef VKMR_HAVE_AVX2_PATH
    if (have_avx2()) return CountLinesAvx2(buf, len);
#endif
    return CountLinesPortable(buf, len);
}

}  // namespace vkmr

extern "C" {

// One-shot form for bindings: packs the whole buffer (final = true).  Returns the
// number of strings, or -1 when a buffer was too small.
__attribute__((visibility("default"))) int64_t vkmr_host_pack_lines(const uint8_t* buf, uint64_t len, uint32_t* data,
                                                                     uint64_t data_capacity_words, vkmr_metadata* meta,
                                                                     uint64_t meta_capacity, uint64_t* words_used,
                                                                     uint64_t* bytes_total)
{
    const vkmr::PackResult r = vkmr::PackLines(buf, len, true, data, 0, data_capacity_words, meta, meta_capacity);
    if (r.consumed != len) return -1;
    if (words_used) *words_used = r.words;
    if (bytes_total) *bytes_total = r.bytes;
    return (int64_t)r.strings;
}

// The same with the portable (memchr per line) splitter forced, and the line counter of the parallel packer's first
// pass in both forms: lets the tests hold the AVX2 forms against the portable ones on the same input.
__attribute__((visibility("default"))) int64_t vkmr_host_pack_lines_portable(const uint8_t* buf, uint64_t len, uint32_t* data,
                                                                              uint64_t data_capacity_words, vkmr_metadata* meta,
                                                                              uint64_t meta_capacity, uint64_t* words_used,
                                                                              uint64_t* bytes_total)
{
    const vkmr::PackResult r = vkmr::PackLinesPortable(buf, len, true, data, 0, data_capacity_words, meta, meta_capacity);
    if (r.consumed != len) return -1;
    if (words_used) *words_used = r.words;
    if (bytes_total) *bytes_total = r.bytes;
    return (int64_t)r.strings;
}

// out[0..4] = strings, words, bytes, empties, too_long; which: 0 = the form the packer uses on this CPU, 1 = portable
__attribute__((visibility("default"))) void vkmr_host_count_lines(const uint8_t* buf, uint64_t len, int which, uint64_t* out)
{
    const vkmr::LineCount c = which ? vkmr::CountLinesPortable(buf, len) : vkmr::CountLines(buf, len);
    out[0] = c.strings; out[1] = c.words; out[2] = c.bytes; out[3] = c.empties; out[4] = c.too_long ? 1 : 0;
}

// A prefix split with capacity limits and `final` unset, as the stream processor calls it: out[0..4] = consumed, strings,
// words, bytes, empties; which as above.
__attribute__((visibility("default"))) void vkmr_host_pack_prefix(const uint8_t* buf, uint64_t len, int final, uint32_t* data, uint64_t first_word,
                                                                  uint64_t data_capacity_words, vkmr_metadata* meta, uint64_t meta_capacity,
                                                                  int which, uint64_t* out)
{
    const vkmr::PackResult r = which ? vkmr::PackLinesPortable(buf, len, final != 0, data, first_word, data_capacity_words, meta, meta_capacity)
                                     : vkmr::PackLines(buf, len, final != 0, data, first_word, data_capacity_words, meta, meta_capacity);
    out[0] = r.consumed; out[1] = r.strings; out[2] = r.words; out[3] = r.bytes; out[4] = r.empties;
}

}  // extern "C"
