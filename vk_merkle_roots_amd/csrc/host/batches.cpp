// batches.cpp -- see batches.hpp.
#include "batches.hpp"

#include <cstring>
#include <iostream>

#include "stream_pack.hpp"

namespace vkmr {

Batch::Batch(Batch&& o) noexcept { *this = std::move(o); }

Batch& Batch::operator=(Batch&& o) noexcept
{
    if (this != &o) {
        Release();
        m_owner = o.m_owner; m_dev = o.m_dev;
        m_data = o.m_data; m_meta = o.m_meta; m_ddata = o.m_ddata; m_dmeta = o.m_dmeta;
        m_cap_words = o.m_cap_words; m_cap_count = o.m_cap_count;
        m_count = o.m_count; m_words = o.m_words; m_bytes = o.m_bytes; m_number = o.m_number;
        o.m_owner = nullptr; o.m_data = nullptr; o.m_meta = nullptr; o.m_ddata = nullptr; o.m_dmeta = nullptr;
        o.m_count = o.m_words = o.m_bytes = 0;
    }
    return *this;
}

void Batch::Release()
{
    if (m_owner && m_data) m_owner->Recycle(*this);
    m_owner = nullptr;
    m_data = nullptr; m_meta = nullptr; m_ddata = nullptr; m_dmeta = nullptr;
    m_count = m_words = m_bytes = 0;
}

bool Batch::Push(const char* p, size_t n)
{
    if (!(*this)) return false;
    const size_t nw = WordCount(n);
    if (m_count + 1 > m_cap_count || m_words + nw > m_cap_words || n > 0xFFFFFFFFull) return false;
    m_meta[m_count].start = (uint32_t)m_words;
    m_meta[m_count].size = (uint32_t)n;
    if (nw) {
        m_data[m_words + nw - 1] = 0u;   // pad bytes of the last word are zero (the kernel masks them anyway)
        std::memcpy(m_data + m_words, p, n);
    }
    m_words += nw;
    m_bytes += n;
    m_count += 1;
    return true;
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
    m_count += r.strings;
    m_words += r.words;
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

Batches::Batches(int dev, size_t data_bytes)
    : m_dev(dev), m_words(data_bytes / 4), m_count(data_bytes / sizeof(vkmr_digest)), m_live(0), m_next(0)
{
    if (m_count == 0) m_count = 1;
}

Batches::~Batches()
{
    for (auto& b : m_free) {
        vkmr_hip_host_free(b.data);
        vkmr_hip_host_free(b.meta);
        vkmr_hip_device_free(m_dev, b.ddata);
        vkmr_hip_device_free(m_dev, b.dmeta);
    }
}

Batch Batches::New()
{
    Batch b;
    Buffers buf = {nullptr, nullptr, nullptr, nullptr};
    if (!m_free.empty()) {
        buf = m_free.back();
        m_free.pop_back();
    } else {
        void *h1 = nullptr, *h2 = nullptr, *d1 = nullptr, *d2 = nullptr;
        const bool ok = vkmr_hip_host_alloc(m_words * 4, &h1) == VKMR_OK &&
                        vkmr_hip_host_alloc(m_count * sizeof(vkmr_metadata), &h2) == VKMR_OK &&
                        vkmr_hip_device_alloc(m_dev, m_words * 4, &d1) == VKMR_OK &&
                        vkmr_hip_device_alloc(m_dev, m_count * sizeof(vkmr_metadata), &d2) == VKMR_OK;
        if (!ok) {
            std::cerr << "Failed to allocate a batch: " << vkmr_hip_last_error() << std::endl;
            vkmr_hip_host_free(h1); vkmr_hip_host_free(h2);
            vkmr_hip_device_free(m_dev, d1); vkmr_hip_device_free(m_dev, d2);
            return b;
        }
        buf = {static_cast<uint32_t*>(h1), static_cast<vkmr_metadata*>(h2), static_cast<uint32_t*>(d1),
               static_cast<vkmr_metadata*>(d2)};
    }
    b.m_owner = this; b.m_dev = m_dev;
    b.m_data = buf.data; b.m_meta = buf.meta; b.m_ddata = buf.ddata; b.m_dmeta = buf.dmeta;
    b.m_cap_words = m_words; b.m_cap_count = m_count;
    b.m_number = m_next++;
    ++m_live;
    return b;
}

void Batches::Recycle(Batch& b)
{
    m_free.push_back({b.m_data, b.m_meta, b.m_ddata, b.m_dmeta});
    --m_live;
}

}  // namespace vkmr
