// batches.cpp -- see batches.hpp.
#include "batches.hpp"

#include <chrono>
#include <cstring>
#include <iostream>
#include <thread>

#include "fork_join.hpp"
#include "timing.hpp"

#include "stream_pack.hpp"
#ifdef VKMR_EXPERIMENTS
#include "vkmr_hip_experiments.h"
#endif

namespace vkmr {

Batch::Batch(Batch&& o) noexcept { *this = std::move(o); }

Batch& Batch::operator=(Batch&& o) noexcept
{
    if (this != &o) {
        Release();
        m_owner = o.m_owner; m_dev = o.m_dev;
        m_data = o.m_data; m_meta = o.m_meta; m_ddata = o.m_ddata; m_dmeta = o.m_dmeta;
        m_sizes = o.m_sizes; m_dsizes = o.m_dsizes; m_dscratch = o.m_dscratch; m_longest = o.m_longest;
        m_dtext = o.m_dtext; m_dsplit = o.m_dsplit; m_dresult = o.m_dresult; m_hresult = o.m_hresult; m_text_bytes = o.m_text_bytes;
        o.m_dtext = nullptr; o.m_dsplit = nullptr; o.m_dresult = nullptr; o.m_hresult = nullptr; o.m_text_bytes = 0;
        m_cap_words = o.m_cap_words; m_cap_count = o.m_cap_count;
        m_count = o.m_count; m_words = o.m_words; m_bytes = o.m_bytes; m_number = o.m_number;
        o.m_owner = nullptr; o.m_data = nullptr; o.m_meta = nullptr; o.m_ddata = nullptr; o.m_dmeta = nullptr;
        o.m_sizes = nullptr; o.m_dsizes = nullptr; o.m_dscratch = nullptr;
        o.m_count = o.m_words = o.m_bytes = o.m_longest = 0;
    }
    return *this;
}

void Batch::Release()
{
    if (m_owner && m_data) m_owner->Recycle(*this);
    m_owner = nullptr;
    m_data = nullptr; m_meta = nullptr; m_ddata = nullptr; m_dmeta = nullptr;
    m_sizes = nullptr; m_dsizes = nullptr; m_dscratch = nullptr;
    m_dtext = nullptr; m_dsplit = nullptr; m_dresult = nullptr; m_hresult = nullptr;
    m_count = m_words = m_bytes = m_longest = m_text_bytes = 0;
}

void Batch::SetText(size_t text_bytes, size_t strings, size_t payload_bytes)
{
    m_text_bytes = text_bytes;
    m_count = strings;
    m_bytes = payload_bytes;
    // the packed words are counted on the device; the host knows them to within 3 bytes per string: what the map launch is
    // told (its mode goes by the average, its bounds by the total) -- never less than what the strings take
    const size_t upper = (payload_bytes + 3 * strings) / 4;
    m_words = upper < m_cap_words ? upper : m_cap_words;
    m_longest = 0xFFFFu;   // the sizes are not on the host: this batch is not described by them
}

void Batch::NoteSizes(size_t first, size_t count)
{
    if (!m_sizes) return;
    size_t longest = m_longest;
    for (size_t i = first; i < first + count; ++i) {
        const uint32_t n = m_meta[i].size;
        m_sizes[i] = (uint16_t)(n < 0xFFFFu ? n : 0xFFFFu);
        longest = n > longest ? n : longest;
    }
    m_longest = longest;
}

bool Batch::Push(const char* p, size_t n)
{
    if (!(*this)) return false;
    const size_t nw = WordCount(n);
    // start is a 32-bit word index and size a 32-bit byte count (vkmr_metadata)
    if (m_count + 1 > m_cap_count || m_words + nw > m_cap_words || n > 0xFFFFFFFFull || m_words > 0xFFFFFFFFull) return false;
    m_meta[m_count].start = (uint32_t)m_words;
    m_meta[m_count].size = (uint32_t)n;
    if (nw) {
        m_data[m_words + nw - 1] = 0u;   // pad bytes of the last word are zero (the kernel masks them anyway)
        std::memcpy(m_data + m_words, p, n);
    }
    NoteSizes(m_count, 1);
    m_words += nw;
    m_bytes += n;
    m_count += 1;
    return true;
}

size_t Batch::PushPacked(const uint32_t* data, const vkmr_metadata* meta, size_t count, size_t max_strings)
{
    if (!(*this) || count == 0) return 0;
    size_t take = count < max_strings ? count : max_strings;
    if (take > m_cap_count - m_count) take = m_cap_count - m_count;
    const size_t first = meta[0].start;
    // the longest prefix whose words fit: strings are consecutive, so the words of [0, k) end where string k - 1 ends
    auto end_of = [&](size_t k) { return (size_t)meta[k - 1].start + WordCount(meta[k - 1].size) - first; };
    size_t lo = 0, hi = take;   // invariant: [0, lo) fits
    while (lo < hi) {
        const size_t mid = (lo + hi + 1) / 2;
        if (m_words + end_of(mid) <= m_cap_words && m_words + end_of(mid) <= 0xFFFFFFFFull) lo = mid; else hi = mid - 1;
    }
    take = lo;
    if (take == 0) return 0;
    const size_t nw = end_of(take);
    std::memcpy(m_data + m_words, data + first, nw * 4);
    size_t bytes = 0;
    for (size_t i = 0; i < take; ++i) {
        m_meta[m_count + i].start = (uint32_t)(m_words + (meta[i].start - first));
        m_meta[m_count + i].size = meta[i].size;
        bytes += meta[i].size;
    }
    NoteSizes(m_count, take);
    m_words += nw;
    m_bytes += bytes;
    m_count += take;
    return take;
}

bool Batch::Push(const std::vector<std::string>& strings)
{
    if (!(*this)) return false;
    size_t nw = 0;
    for (const auto& s : strings) nw += WordCount(s.size());
    if (m_count + strings.size() > m_cap_count || m_words + nw > m_cap_words) return false;   // all or nothing
    for (const auto& s : strings)
        if (!Push(s.data(), s.size())) return false;
    return true;
}

PackResult Batch::PushLines(const char* buf, size_t len, bool final, size_t max_strings)
{
    PackResult r = {0, 0, 0, 0, 0};
    if (!(*this)) return r;
    size_t room = m_cap_count - m_count;
    if (room > max_strings) room = max_strings;
    r = PackLines(reinterpret_cast<const uint8_t*>(buf), len, final, m_data, m_words, m_cap_words, m_meta + m_count, room);
    NoteSizes(m_count, r.strings);
    m_count += r.strings;
    m_words += r.words;
    m_bytes += r.bytes;
    return r;
}

PackResult Batch::PushLinesParallel(const char* buf, size_t len, bool final, size_t max_strings, ForkJoin& pool, double words_per_byte)
{
    PackResult r = {0, 0, 0, 0, 0};
    const unsigned threads = pool.Width();
    if (!(*this) || threads < 2 || len < (1u << 20)) return r;
    const uint8_t* b = reinterpret_cast<const uint8_t*>(buf);
    // only whole lines: a span that is not final stops at its last '\n'
    size_t usable = len;
    if (!final) {
        const void* nl = memrchr(b, '\n', len);
        usable = nl ? (size_t)(static_cast<const uint8_t*>(nl) - b) + 1 : 0;
    }
    if (usable < (1u << 20)) return r;
    // What will not fit is not worth indexing: a span is cut to what the room left in the batch is likely to hold (the
    // caller's words-per-byte figure of the stream so far, with 2 % to spare).  Without this a 64 MiB batch fed 32 MiB
    // spans indexed every other span twice (profiles/r03_frontend_phases.txt: 172 index passes for 65 spans).
    if (words_per_byte > 0.0) {
        const double fits = (double)(m_cap_words - m_words) / words_per_byte * 0.98;
        if (fits < (double)usable) {
            if (fits < (double)(1u << 20)) return r;   // (nearly) full: the caller tops it up line by line, or sends it off
            const void* nl = memrchr(b, '\n', (size_t)fits);
            if (!nl) return r;
            usable = (size_t)(static_cast<const uint8_t*>(nl) - b) + 1;
        }
    }
    // parts are indexed with 32-bit offsets, and the index of a part is four times its size at worst (all lines empty):
    // a very large span is taken in several calls (the caller comes back with the rest)
    const size_t most = (size_t)threads << 26;
    if (usable > most) {
        const void* nl = memrchr(b, '\n', most);
        if (!nl) return r;   // one line of more than 64 MiB x threads: the serial form deals with it
        usable = (size_t)(static_cast<const uint8_t*>(nl) - b) + 1;
    }
    // 1. parts that end after a '\n'
    struct Part { size_t lo, hi; LineCount c; };
    std::vector<Part> parts;
    const size_t per = usable / threads;
    size_t lo = 0;
    for (unsigned t = 0; t < threads && lo < usable; ++t) {
        size_t hi = (t + 1 == threads) ? usable : per * (t + 1);
        if (hi < lo) hi = lo;
        if (hi < usable) {
            const void* nl = memchr(b + hi, '\n', usable - hi);
            hi = nl ? (size_t)(static_cast<const uint8_t*>(nl) - b) + 1 : usable;
        }
        if (hi > lo) {
            if (hi - lo >= 0xFFFFFF00ull) return r;   // (a single line that long)
            parts.push_back({lo, hi, {0, 0, 0, 0, false}});
        }
        lo = hi;
    }
    // 2. index every part: where its lines end, and what it will append
    static thread_local std::vector<LineIndex> kept;   // from span to span, one per part
    if (kept.size() < parts.size()) kept.resize(parts.size());
    LineIndex* const index = kept.data();   // (a thread_local named inside the tasks would be each worker's own, empty, vector)
    {
        timing::Scope ts(timing::INDEX);
        pool.Run((unsigned)parts.size(), [&](unsigned t) { parts[t].c = IndexLines(b + parts[t].lo, parts[t].hi - parts[t].lo, index + t); });
    }
    for (size_t t = 0; t < parts.size(); ++t)
        if (!index[t].ends) return r;   // out of memory for an index: the serial form needs none
    // 3. leading parts that fit entirely
    size_t room = m_cap_count - m_count;
    if (room > max_strings) room = max_strings;
    size_t k = 0, words = 0, strings = 0;
    std::vector<size_t> w0, c0;
    for (; k < parts.size(); ++k) {
        const LineCount& c = parts[k].c;
        if (strings + c.strings > room || m_words + words + c.words > m_cap_words || m_words + words + c.words > 0xFFFFFFFFull)
            break;
        w0.push_back(m_words + words);
        c0.push_back(m_count + strings);
        words += c.words;
        strings += c.strings;
    }
    if (k == 0) return r;
    // 4. pack them at their prefix offsets, each within its own words; with ordinary or streaming stores, as the pool's tuner says
    {
        timing::Scope ts(timing::PACK);
        const bool streaming = m_owner ? m_owner->Tuner().Next() : false;
        const auto t0 = std::chrono::steady_clock::now();
        pool.Run((unsigned)k, [&](unsigned t) {
            PackIndexed(b + parts[t].lo, parts[t].hi - parts[t].lo, index[t], m_data, w0[t], w0[t] + parts[t].c.words, m_meta + c0[t],
                        m_sizes ? m_sizes + c0[t] : nullptr, streaming);
        });
        if (m_owner) m_owner->Tuner().Report(streaming, words * 4, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    for (size_t t = 0; t < k; ++t) {
        r.bytes += parts[t].c.bytes;
        r.empties += parts[t].c.empties;
        if (parts[t].c.longest > m_longest) m_longest = parts[t].c.longest;
    }
    r.consumed = parts[k - 1].hi;
    r.strings = strings;
    r.words = words;
    m_count += strings;
    m_words += words;
    m_bytes += r.bytes;
    return r;
}

void Batch::Pop(size_t count)
{
    while (count-- && m_count) {
        --m_count;
        m_words = m_meta[m_count].start;
        m_bytes -= m_meta[m_count].size;
    }
}

Batches::Batches(int dev, size_t data_bytes, bool device_split, int pack_stream)
    : m_dev(dev), m_device_split(device_split), m_tuner(pack_stream), m_words(data_bytes / 4), m_count(data_bytes / sizeof(vkmr_digest)), m_live(0), m_next(0)
{
    if (m_words > 0xFFFFFFFFull) m_words = 0xFFFFFFFFull;   // vkmr_metadata::start is a 32-bit word index
    if (m_count == 0) m_count = 1;
}

void Batches::Free(Buffers& b)
{
    vkmr_hip_host_free(b.data);
    vkmr_hip_host_free(b.meta);
    vkmr_hip_host_free(b.sizes);
    vkmr_hip_device_free(m_dev, b.ddata);
    vkmr_hip_device_free(m_dev, b.dmeta);
    vkmr_hip_device_free(m_dev, b.dsizes);
    vkmr_hip_device_free(m_dev, b.dscratch);
    vkmr_hip_device_free(m_dev, b.dtext);
    vkmr_hip_device_free(m_dev, b.dsplit);
    vkmr_hip_device_free(m_dev, b.dresult);
    vkmr_hip_host_free(b.hresult);
}

Batches::~Batches()
{
    JoinPrefetch();
    for (auto& b : m_free) Free(b);
}

void Batches::JoinPrefetch()
{
    if (!m_prefetch.joinable()) return;
    m_stop = true;   // at most the allocation in progress is waited for
    m_prefetch.join();
    m_stop = false;
}

bool Batches::Allocate(size_t words, size_t count, Buffers* out)
{
    void *h1 = nullptr, *h2 = nullptr, *h3 = nullptr, *d1 = nullptr, *d2 = nullptr, *d3 = nullptr, *d4 = nullptr;
    const size_t count32 = count > 0xFFFFFFFFull ? 0xFFFFFFFFull : count;
    const bool ok = vkmr_hip_host_alloc(words * 4, &h1) == VKMR_OK && vkmr_hip_host_alloc(count * sizeof(vkmr_metadata), &h2) == VKMR_OK &&
                    vkmr_hip_host_alloc(count * sizeof(uint16_t), &h3) == VKMR_OK &&
                    vkmr_hip_device_alloc(m_dev, words * 4, &d1) == VKMR_OK &&
                    vkmr_hip_device_alloc(m_dev, count * sizeof(vkmr_metadata), &d2) == VKMR_OK &&
                    vkmr_hip_device_alloc(m_dev, count * sizeof(uint16_t), &d3) == VKMR_OK &&
                    vkmr_hip_device_alloc(m_dev, vkmr_hip_sizes_scratch_bytes((uint32_t)count32), &d4) == VKMR_OK;
    if (!ok) {   // the caller waits for a mapping to retire and tries again, or reports the failure
        vkmr_hip_host_free(h1); vkmr_hip_host_free(h2); vkmr_hip_host_free(h3);
        vkmr_hip_device_free(m_dev, d1); vkmr_hip_device_free(m_dev, d2); vkmr_hip_device_free(m_dev, d3); vkmr_hip_device_free(m_dev, d4);
        return false;
    }
    *out = {static_cast<uint32_t*>(h1), static_cast<vkmr_metadata*>(h2), static_cast<uint32_t*>(d1), static_cast<vkmr_metadata*>(d2), words, count,
            static_cast<uint16_t*>(h3), static_cast<uint16_t*>(d3), d4, nullptr, nullptr, nullptr, nullptr};
#ifdef VKMR_EXPERIMENTS
    if (m_device_split && words * 4 < 0xFFFFFFE0ull) {   // the splitter's text area, scratch and result words; without them the batch simply cannot hold text
        void *t1 = nullptr, *t2 = nullptr, *t3 = nullptr, *t4 = nullptr;
        const bool ok2 = vkmr_hip_device_alloc(m_dev, words * 4 + 64, &t1) == VKMR_OK &&
                         vkmr_hip_device_alloc(m_dev, vkmr_hip_split_scratch_bytes((uint32_t)(words * 4), (uint32_t)count32), &t2) == VKMR_OK &&
                         vkmr_hip_device_alloc(m_dev, 16, &t3) == VKMR_OK && vkmr_hip_host_alloc(16, &t4) == VKMR_OK;
        if (ok2) {
            out->dtext = static_cast<uint8_t*>(t1); out->dsplit = t2; out->dresult = static_cast<uint32_t*>(t3); out->hresult = static_cast<uint32_t*>(t4);
        } else {
            vkmr_hip_device_free(m_dev, t1); vkmr_hip_device_free(m_dev, t2); vkmr_hip_device_free(m_dev, t3); vkmr_hip_host_free(t4);
        }
    }
#endif
    return true;
}

void Batches::Prefetch(size_t n)
{
    JoinPrefetch();
    if (n == 0) return;
    const size_t words = m_words, count = m_count;
    {
        std::lock_guard<std::mutex> lock(m_mu);
        m_pending = n;
    }
    m_prefetch = std::thread([this, n, words, count] {
        for (size_t i = 0; i < n; ++i) {
            Buffers buf;
            const bool ok = !m_stop && Allocate(words, count, &buf);
            std::lock_guard<std::mutex> lock(m_mu);
            if (ok) {
                m_free.push_back(buf);
                ++m_allocations;
                --m_pending;
            } else {
                m_pending = 0;   // out of memory: the owner allocates (and reports) for itself
            }
            m_cv.notify_all();
            if (!ok) return;
        }
    });
}

void Batches::Reshape(size_t data_bytes, size_t meta_count)
{
    size_t words = data_bytes / 4;
    if (words > 0xFFFFFFFFull) words = 0xFFFFFFFFull;
    if (words == 0 || meta_count == 0 || (words == m_words && meta_count == m_count)) return;
    JoinPrefetch();
    m_words = words;
    m_count = meta_count;
    for (auto& b : m_free) Free(b);   // idle buffers of the old shape
    m_free.clear();
}

Batch Batches::New()
{
    Batch b;
    Buffers buf = {nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool have = false;
    {
        std::unique_lock<std::mutex> lock(m_mu);
        m_cv.wait(lock, [&] { return !m_free.empty() || m_pending == 0; });   // a prefetched batch is on its way
        if (!m_free.empty()) {
            buf = m_free.back();
            m_free.pop_back();
            have = true;
        }
    }
    if (!have) {
        if (!Allocate(m_words, m_count, &buf)) return b;
        std::lock_guard<std::mutex> lock(m_mu);
        ++m_allocations;
    }
    b.m_owner = this; b.m_dev = m_dev;
    b.m_data = buf.data; b.m_meta = buf.meta; b.m_ddata = buf.ddata; b.m_dmeta = buf.dmeta;
    b.m_sizes = buf.sizes; b.m_dsizes = buf.dsizes; b.m_dscratch = buf.dscratch;
    b.m_dtext = buf.dtext; b.m_dsplit = buf.dsplit; b.m_dresult = buf.dresult; b.m_hresult = buf.hresult;
    b.m_cap_words = buf.words; b.m_cap_count = buf.count;
    b.m_number = m_next++;
    ++m_live;
    return b;
}

void Batches::Recycle(Batch& b)
{
    Buffers buf = {b.m_data, b.m_meta, b.m_ddata, b.m_dmeta, b.m_cap_words, b.m_cap_count, b.m_sizes, b.m_dsizes, b.m_dscratch,
                   b.m_dtext, b.m_dsplit, b.m_dresult, b.m_hresult};
    if (buf.words == m_words && buf.count == m_count) {
        std::lock_guard<std::mutex> lock(m_mu);
        m_free.push_back(buf);
    } else {
        Free(buf);   // shape changed since this batch was handed out
    }
    --m_live;
}

}  // namespace vkmr
