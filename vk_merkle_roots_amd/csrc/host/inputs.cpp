// inputs.cpp -- see inputs.hpp.
#include "inputs.hpp"

#include <cstdlib>
#include <cstring>

#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace vkmr {

Input::Input(FILE* fp, bool owner)
    : m_fp(fp), m_owner(owner), m_eof(fp == nullptr), m_size(0), m_count(0), m_buf(1 << 20), m_pos(0), m_end(0)
{
    struct stat st;
    if (fp && fstat(fileno(fp), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        const off_t at = lseek(fileno(fp), 0, SEEK_CUR);
        if (at == 0) {
            void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fp), 0);
            if (p != MAP_FAILED) {
                madvise(p, (size_t)st.st_size, MADV_SEQUENTIAL);
                m_map = static_cast<const char*>(p);
                m_map_len = (size_t)st.st_size;
            }
        }
    }
}

Input::Input(const std::string& path) : Input(fopen(path.c_str(), "r"), true) {}

Input::~Input()
{
    if (m_map) munmap(const_cast<char*>(m_map), m_map_len);
    if (m_owner && m_fp) fclose(m_fp);
}

size_t Input::ReadSome(char* dst, size_t n)
{
    if (m_map) {
        const size_t left = m_map_len - m_map_pos;
        const size_t take = n < left ? n : left;
        memcpy(dst, m_map + m_map_pos, take);
        m_map_pos += take;
        return take;
    }
    return m_fp ? fread(dst, 1, n, m_fp) : 0;
}

bool Input::Fill()
{
    m_pos = 0;
    m_end = ReadSome(m_buf.data(), m_buf.size());
    return m_end > 0;
}

bool Input::GetView(const char** p, size_t* n)
{
    m_carry.clear();
    bool carried = false;
    for (;;) {
        if (m_pos == m_end && !Fill()) {
            m_eof = true;   // the read that hit EOF ends the current (possibly empty) line
            break;
        }
        const char* base = m_buf.data() + m_pos;
        const char* nl = static_cast<const char*>(memchr(base, '\n', m_end - m_pos));
        if (nl) {
            const size_t len = (size_t)(nl - base);
            m_pos += len + 1;
            if (!carried) {
                *p = base;
                *n = len;
                m_size += len;
                m_count += len ? 1 : 0;
                return true;
            }
            m_carry.append(base, len);
            break;
        }
        m_carry.append(base, m_end - m_pos);   // line continues in the next block
        carried = true;
        m_pos = m_end;
    }
    *p = m_carry.data();
    *n = m_carry.size();
    m_size += m_carry.size();
    m_count += m_carry.empty() ? 0 : 1;
    return true;
}

bool Input::GetBlock(const char** p, size_t* n, bool* final)
{
    if (m_map && m_pos == m_end) {
        // mapped file, nothing buffered: hand out the next span of the mapping, cut after a '\n'
        static const size_t span = [] { const char* e = getenv("VKMR_INPUT_SPAN_MB"); const long v = e ? atol(e) : 32; return (size_t)(v < 1 ? 1 : v) << 20; }();
        const size_t left = m_map_len - m_map_pos;
        size_t take = left;
        if (left > span) {
            const void* nl = memrchr(m_map + m_map_pos, '\n', span);
            if (nl) {
                take = (size_t)(static_cast<const char*>(nl) - (m_map + m_map_pos)) + 1;
            } else {   // a line longer than the span: up to its end
                const void* fwd = memchr(m_map + m_map_pos + span, '\n', left - span);
                take = fwd ? (size_t)(static_cast<const char*>(fwd) - (m_map + m_map_pos)) + 1 : left;
            }
        }
        *p = m_map + m_map_pos;
        *n = take;
        m_map_pos += take;
        *final = (m_map_pos == m_map_len);
        if (*final) m_eof = true;
        return true;
    }
    // keep the unconsumed tail (an incomplete line) at the front, then read more behind it
    if (m_pos > 0 && m_pos < m_end) memmove(m_buf.data(), m_buf.data() + m_pos, m_end - m_pos);
    m_end -= m_pos;
    m_pos = 0;
    if (m_buf.size() < ((size_t)16 << 20)) m_buf.resize((size_t)16 << 20);   // bulk reads: large blocks
    for (;;) {
        if (m_end == m_buf.size()) m_buf.resize(m_buf.size() * 2);   // a line longer than the buffer
        const size_t got = ReadSome(m_buf.data() + m_end, m_buf.size() - m_end);
        m_end += got;
        if (got == 0) {   // end of stream: everything left is the final span
            *p = m_buf.data();
            *n = m_end;
            *final = true;
            m_pos = m_end;
            m_eof = true;
            return true;
        }
        // last '\n' in the buffer (search backwards from the end)
        const void* nl = memrchr(m_buf.data(), '\n', m_end);
        if (nl) {
            const size_t cut = (size_t)(static_cast<const char*>(nl) - m_buf.data()) + 1;
            *p = m_buf.data();
            *n = cut;
            *final = false;
            m_pos = cut;
            return true;
        }
    }
}

std::string Input::Get()
{
    const char* p = nullptr;
    size_t n = 0;
    GetView(&p, &n);
    return std::string(p, n);
}

}  // namespace vkmr
