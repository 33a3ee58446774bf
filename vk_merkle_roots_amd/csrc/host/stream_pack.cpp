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
    LineCount c = {0, 0, 0, 0, false, 0};
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

// ---- indexed two-pass form, portable ------------------------------------------------------------------------

LineIndex::~LineIndex() { free(ends); }

void LineIndex::Reserve(size_t lines)
{
    count = 0;
    if (lines + 16 <= cap) return;
    free(ends);
    cap = lines + 16;
    ends = static_cast<uint32_t*>(malloc(cap * sizeof(uint32_t)));   // not touched until written: a part of long lines uses a page of it
    if (!ends) cap = 0;
}

namespace {

// What the lines of an index add up to.  Line i spans (ends[i-1], ends[i]); the first starts at 0.
LineCount totals_of(const LineIndex& ix)
{
    LineCount c = {0, 0, 0, 0, false, 0};
    const uint32_t* e = ix.ends;
    const size_t k = ix.count;
    if (k == 0) return c;
    uint64_t strings = e[0] != 0, words = ((uint64_t)e[0] + 3u) >> 2;
    uint32_t longest = e[0];
    for (size_t i = 1; i < k; ++i) {
        const uint32_t n = e[i] - e[i - 1] - 1u;
        strings += n != 0;
        words += (n + 3u) >> 2;
        longest = n > longest ? n : longest;
    }
    c.longest = longest;
    c.strings = strings;
    c.words = words;
    c.bytes = (uint64_t)e[k - 1] + 1u - k;   // every line but its newline
    c.empties = k - strings;
    return c;
}

}  // namespace

LineCount IndexLinesPortable(const uint8_t* buf, size_t len, LineIndex* ix)
{
    ix->Reserve(len + 1);
    if (!ix->ends) { ix->count = 0; return LineCount(); }   // no memory for the index: the callers test `ends` and take the serial form
    uint32_t* o = ix->ends;
    size_t pos = 0;
    while (pos < len) {
        const uint8_t* nl = static_cast<const uint8_t*>(memchr(buf + pos, '\n', len - pos));
        const size_t end = nl ? (size_t)(nl - buf) : len;
        *o++ = (uint32_t)end;
        pos = end + 1;
    }
    ix->count = (size_t)(o - ix->ends);
    return totals_of(*ix);
}

void PackIndexedPortable(const uint8_t* buf, size_t len, const LineIndex& ix, uint32_t* data, uint64_t first_word, uint64_t end_word,
                         vkmr_metadata* meta, uint16_t* sizes)
{
    (void)len; (void)end_word;
    uint64_t w = first_word;
    size_t s = 0;
    uint32_t start = 0;
    for (size_t i = 0; i < ix.count; ++i) {
        const uint32_t n = ix.ends[i] - start;
        if (n) {
            const uint32_t nw = (n + 3u) >> 2;
            meta[s].start = (uint32_t)w;
            meta[s].size = n;
            if (sizes) sizes[s] = (uint16_t)(n < 0xFFFFu ? n : 0xFFFFu);
            ++s;
            data[w + nw - 1] = 0u;
            memcpy(data + w, buf + start, n);
            w += nw;
        }
        start = ix.ends[i] + 1u;
    }
}

TextCount CopyAndCountLinesPortable(const uint8_t* buf, size_t len, uint8_t* dst, bool after_newline)
{
    TextCount c = {0, 0};
    if (len) memcpy(dst, buf, len);
    bool prev = after_newline;
    for (size_t i = 0; i < len; ++i) {
        const bool nl = buf[i] == '\n';
        c.newlines += nl;
        c.empties += nl && prev;
        prev = nl;
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
    LineCount c = {0, 0, 0, 0, false, 0};
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

// Pass 1 of the indexed form.  The walk over the set bits of a block's newline mask is what the line-at-a-time loops
// above mispredict on (0 to 3 newlines per 64 bytes of a rndm stream, at random): here four positions are written
// whatever the count -- tzcnt of an exhausted mask gives 64, a value that the next block overwrites -- and the output
// advances by the population count.
__attribute__((target("avx2,bmi,popcnt"))) LineCount IndexLinesAvx2(const uint8_t* buf, size_t len, LineIndex* ix)
{
    ix->Reserve(len + 1);
    if (!ix->ends) { ix->count = 0; return LineCount(); }   // no memory for the index: the callers test `ends` and take the serial form
    uint32_t* o = ix->ends;
    size_t blk = 0;
    for (; blk + 64 <= len; blk += 64) {
        uint64_t m = newline_mask64(buf + blk);
        const unsigned cnt = (unsigned)__builtin_popcountll(m);
        const uint32_t base = (uint32_t)blk;
        o[0] = base + (uint32_t)_tzcnt_u64(m); m = _blsr_u64(m);
        o[1] = base + (uint32_t)_tzcnt_u64(m); m = _blsr_u64(m);
        o[2] = base + (uint32_t)_tzcnt_u64(m); m = _blsr_u64(m);
        o[3] = base + (uint32_t)_tzcnt_u64(m); m = _blsr_u64(m);
        if (__builtin_expect(cnt > 4, 0)) {
            for (unsigned k = 4; k < cnt; ++k) {
                o[k] = base + (uint32_t)_tzcnt_u64(m);
                m = _blsr_u64(m);
            }
        }
        o += cnt;
    }
    for (size_t i = blk; i < len; ++i)
        if (buf[i] == '\n') *o++ = (uint32_t)i;
    if (len > 0 && buf[len - 1] != '\n') *o++ = (uint32_t)len;   // the unterminated last line
    ix->count = (size_t)(o - ix->ends);
    return totals_of(*ix);
}

// Pass 2.  A line of up to 128 bytes is moved as four 32-byte vectors whatever its length -- what lands behind its end
// is overwritten by the next line, and the bytes of its last word beyond its end are cleared afterwards -- as long as
// that stays inside this part's input and this part's words; longer lines, and the last few of a part, go through memcpy.
__attribute__((target("avx2"))) void PackIndexedAvx2(const uint8_t* buf, size_t len, const LineIndex& ix, uint32_t* data, uint64_t first_word,
                                                     uint64_t end_word, vkmr_metadata* meta, uint16_t* sizes)
{
    uint64_t w = first_word;
    vkmr_metadata* mo = meta;
    uint16_t scrap;
    uint16_t* so = sizes ? sizes : &scrap;
    const size_t so_step = sizes ? 1 : 0;
    uint32_t start = 0;
    const uint32_t* ends = ix.ends;
    for (size_t i = 0; i < ix.count; ++i) {
        const uint32_t e = ends[i];
        const uint32_t n = e - start;
        if (n) {
            const uint32_t nw = (n + 3u) >> 2;
            mo->start = (uint32_t)w;
            mo->size = n;
            ++mo;
            *so = (uint16_t)(n < 0xFFFFu ? n : 0xFFFFu);
            so += so_step;
            const uint8_t* src = buf + start;
            uint8_t* dst = reinterpret_cast<uint8_t*>(data + w);
            if (n <= 128u && (size_t)start + 128u <= len && w + 32u <= end_word) {
                const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src));
                const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 32));
                const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 64));
                const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 96));
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst), a);
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + 32), b);
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + 64), c);
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + 96), d);
                data[w + nw - 1] &= 0xFFFFFFFFu >> (8u * ((0u - n) & 3u));
            } else {
                data[w + nw - 1] = 0u;
                memcpy(dst, src, n);
            }
            w += nw;
        }
        start = e + 1u;
    }
}

// 64 bytes at a time: two vector loads, two stores, the newline mask; an empty line is a newline bit whose lower neighbour
// (the last bit of the block before, for bit 0) is one too.
__attribute__((target("avx2,popcnt"))) TextCount CopyAndCountLinesAvx2(const uint8_t* buf, size_t len, uint8_t* dst, bool after_newline)
{
    TextCount c = {0, 0};
    const __m256i nl = _mm256_set1_epi8('\n');
    uint64_t carry = after_newline ? 1u : 0u;
    size_t i = 0;
    for (; i + 64 <= len; i += 64) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(buf + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(buf + i + 32));
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + i), a);
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + i + 32), b);
        const uint64_t m = (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, nl)) |
                           ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, nl)) << 32);
        c.newlines += (uint64_t)__builtin_popcountll(m);
        c.empties += (uint64_t)__builtin_popcountll(m & ((m << 1) | carry));
        carry = m >> 63;
    }
    const TextCount t = CopyAndCountLinesPortable(buf + i, len - i, dst + i, carry != 0);
    c.newlines += t.newlines;
    c.empties += t.empties;
    return c;
}

// [done, upto) of the destination from the window whose first byte is destination byte `lo` (a multiple of 64): ordinary
// stores up to the first line boundary and for the last, partial, line; streaming stores for the whole lines between.
__attribute__((target("avx2"))) uint64_t stream_out(uint8_t* dst, const uint8_t* win, uint64_t lo, uint64_t done, uint64_t upto)
{
    if (upto <= done) return done;
    uint64_t a = done;
    const uint64_t head_end = (a + 63u) & ~(uint64_t)63;
    if (a < head_end) {
        const uint64_t e = head_end < upto ? head_end : upto;
        memcpy(dst + a, win + (a - lo), (size_t)(e - a));
        a = e;
    }
    for (; a + 64u <= upto; a += 64u) {
        const __m256i x = _mm256_load_si256(reinterpret_cast<const __m256i*>(win + (a - lo)));
        const __m256i y = _mm256_load_si256(reinterpret_cast<const __m256i*>(win + (a - lo) + 32));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + a), x);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + a + 32), y);
    }
    if (a < upto) memcpy(dst + a, win + (a - lo), (size_t)(upto - a));
    return upto;
}

// Pass 2 with non-temporal stores.  The destination is written once and read next by the copy engine, so a core that
// brings every destination line into its cache first (read for ownership) moves a third more memory than it has to --
// unless the batch then stays in the last-level cache and the copy engine reads it there, which is what happens on a quiet
// host and not on a busy one (profiles/r03_frontend_streaming_stores.txt): the caller tries both and keeps the faster
// (PackTuner).  The lines are assembled in a small window on the stack -- where the four-vector copy may run past a
// line's end freely -- and the window's completed 64-byte lines go to the destination with streaming stores; the part's
// first and last, partial, lines are written the ordinary way (their other bytes are another thread's).
__attribute__((target("avx2"))) void PackIndexedAvx2Stream(const uint8_t* buf, size_t len, const LineIndex& ix, uint32_t* data, uint64_t first_word,
                                                           uint64_t end_word, vkmr_metadata* meta, uint16_t* sizes)
{
    (void)end_word;
    constexpr size_t WIN = 8192;
    alignas(64) uint8_t win[WIN + 256];
    // destination offsets are counted from a 64-byte-aligned address at or below `data` (streaming stores need aligned lines)
    const uint64_t skew = reinterpret_cast<uintptr_t>(data) & 63u;
    uint8_t* const dst = reinterpret_cast<uint8_t*>(data) - skew;
    const uint64_t first_b = first_word * 4u + skew;
    uint64_t lo = first_b & ~(uint64_t)63;   // destination offset of win[0]
    uint64_t done = first_b;                 // destination bytes below this are written
    uint64_t w = first_word;
    vkmr_metadata* mo = meta;
    uint16_t scrap;
    uint16_t* so = sizes ? sizes : &scrap;
    const size_t so_step = sizes ? 1 : 0;
    uint32_t start = 0;
    const uint32_t* ends = ix.ends;
    for (size_t i = 0; i < ix.count; ++i) {
        const uint32_t e = ends[i];
        const uint32_t n = e - start;
        if (n) {
            const uint32_t nw = (n + 3u) >> 2;
            mo->start = (uint32_t)w;
            mo->size = n;
            ++mo;
            *so = (uint16_t)(n < 0xFFFFu ? n : 0xFFFFu);
            so += so_step;
            const uint64_t pos = w * 4u + skew;
            const uint8_t* src = buf + start;
            if (n <= 128u && (size_t)start + 128u <= len) {
                if (pos + 128u > lo + WIN) {   // the window is full: its whole lines go out, what is left of the current line moves to the front
                    const uint64_t keep = pos & ~(uint64_t)63;
                    done = stream_out(dst, win, lo, done, keep < done ? done : keep);
                    if (keep > lo) {
                        memmove(win, win + (keep - lo), (size_t)(pos - keep));
                        lo = keep;
                    }
                }
                uint8_t* d = win + (pos - lo);
                const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src));
                const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 32));
                const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 64));
                const __m256i dd = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 96));
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(d), a);
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(d + 32), b);
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(d + 64), c);
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(d + 96), dd);
                uint32_t last;
                memcpy(&last, d + 4u * (nw - 1u), 4);
                last &= 0xFFFFFFFFu >> (8u * ((0u - n) & 3u));
                memcpy(d + 4u * (nw - 1u), &last, 4);
            } else {   // a long line, or one of the part's last: the window goes out, the line is copied directly
                done = stream_out(dst, win, lo, done, pos);
                data[w + nw - 1] = 0u;
                memcpy(dst + pos, src, n);
                done = pos + 4u * (uint64_t)nw;
                lo = done & ~(uint64_t)63;
            }
            w += nw;
        }
        start = e + 1u;
    }
    done = stream_out(dst, win, lo, done, w * 4u + skew);
    _mm_sfence();
}

bool have_avx2()
{
    static const bool yes = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi") && __builtin_cpu_supports("popcnt") && !getenv("VKMR_NO_AVX2");
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

TextCount CopyAndCountLines(const uint8_t* buf, size_t len, uint8_t* dst, bool after_newline)
{
#ifdef VKMR_HAVE_AVX2_PATH
    if (have_avx2()) return CopyAndCountLinesAvx2(buf, len, dst, after_newline);
#endif
    return CopyAndCountLinesPortable(buf, len, dst, after_newline);
}

LineCount IndexLines(const uint8_t* buf, size_t len, LineIndex* ix)
{
#ifdef VKMR_HAVE_AVX2_PATH
    if (have_avx2()) return IndexLinesAvx2(buf, len, ix);
#endif
    return IndexLinesPortable(buf, len, ix);
}

void PackIndexed(const uint8_t* buf, size_t len, const LineIndex& ix, uint32_t* data, uint64_t first_word, uint64_t end_word, vkmr_metadata* meta,
                 uint16_t* sizes, bool streaming)
{
#ifdef VKMR_HAVE_AVX2_PATH
    if (have_avx2())
        return streaming ? PackIndexedAvx2Stream(buf, len, ix, data, first_word, end_word, meta, sizes)
                         : PackIndexedAvx2(buf, len, ix, data, first_word, end_word, meta, sizes);
#endif
    PackIndexedPortable(buf, len, ix, data, first_word, end_word, meta, sizes);
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

// The indexed two-pass form on one buffer (every line, the last one terminated or not), placed at data[first_word ...):
// which as above.  Returns the number of strings, or -1 when a buffer was too small; out[0..3] = words, bytes, empties,
// lines indexed.
__attribute__((visibility("default"))) int64_t vkmr_host_pack_indexed(const uint8_t* buf, uint64_t len, uint32_t* data, uint64_t first_word,
                                                                       uint64_t data_capacity_words, vkmr_metadata* meta, uint64_t meta_capacity,
                                                                       int which, uint64_t* out)
{
    if (len >= 0xFFFFFF00ull) return -1;
    vkmr::LineIndex ix;
    const vkmr::LineCount c = which == 1 ? vkmr::IndexLinesPortable(buf, len, &ix) : vkmr::IndexLines(buf, len, &ix);
    if (out) { out[0] = c.words; out[1] = c.bytes; out[2] = c.empties; out[3] = ix.count; }
    if (!ix.ends || first_word + c.words > data_capacity_words || c.strings > meta_capacity) return -1;
    if (which == 1) vkmr::PackIndexedPortable(buf, len, ix, data, first_word, first_word + c.words, meta);
    else vkmr::PackIndexed(buf, len, ix, data, first_word, first_word + c.words, meta, nullptr, which == 2);   // 2: the streaming-store form
    return (int64_t)c.strings;
}

// CopyAndCountLines on one buffer: out[0..1] = newlines, empties; which as above.
__attribute__((visibility("default"))) void vkmr_host_copy_and_count(const uint8_t* buf, uint64_t len, uint8_t* dst, int after_newline, int which, uint64_t* out)
{
    const vkmr::TextCount c = which ? vkmr::CopyAndCountLinesPortable(buf, len, dst, after_newline != 0) : vkmr::CopyAndCountLines(buf, len, dst, after_newline != 0);
    out[0] = c.newlines;
    out[1] = c.empties;
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
