// inputs.cpp -- see inputs.hpp.
#include "inputs.hpp"

#include <cstring>

namespace vkmr {

Input::Input(FILE* fp, bool owner)
    : m_fp(fp), m_owner(owner), m_eof(fp == nullptr), m_size(0), m_count(0), m_buf(1 << 20), m_pos(0), m_end(0)
{
}

Input::Input(const std::string& path) : Input(fopen(path.c_str(), "r"), true) {}

Input::~Input()
{
    if (m_owner && m_fp) fclose(m_fp);
}

bool Input::Fill()
{
    m_pos = 0;
    m_end = m_fp ? fread(m_buf.data(), 1, m_buf.size(), m_fp) : 0;
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
    // keep the unconsumed tail (an incomplete line) at the front, then read more behind it
    if (m_pos > 0 && m_pos < m_end) memmove(m_buf.data(), m_buf.data() + m_pos, m_end - m_pos);
    m_end -= m_pos;
    m_pos = 0;
    for (;;) {
        if (m_end == m_buf.size()) m_buf.resize(m_buf.size() * 2);   // a line longer than the buffer
        const size_t got = m_fp ? fread(m_buf.data() + m_end, 1, m_buf.size() - m_end, m_fp) : 0;
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
        size_t cut = m_end;
        while (cut > 0 && m_buf[cut - 1] != '\n') --cut;
        if (cut > 0) {
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
