// stream_pack.cpp -- see stream_pack.hpp.
#include "stream_pack.hpp"

#include <cstring>

namespace vkmr {

PackResult PackLines(const uint8_t* buf, size_t len, bool final, uint32_t* data, uint64_t first_word,
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

LineCount CountLines(const uint8_t* buf, size_t len)
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

}  // extern "C"
