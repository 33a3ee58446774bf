// inputs.hpp -- line reader for the input stream.
//
// Same contract as the reference's vkmr::Input (src/vkmr/Inputs.h:20-47,
// src/vkmr/Inputs.cpp:52-101): '\n' or end of file ends a line, '\r' is kept, Has() is
// "not at end of file yet" -- so a stream that ends in '\n' yields one final empty
// string, which the caller skips with a warning (src/vkmr/Vkmr.cpp:40-43).  The
// reference reads with one fgetc + append per byte; this reader pulls 1 MiB blocks
// with fread and splits with memchr.
#pragma once
#include <cstdio>
#include <string>
#include <vector>

namespace vkmr {

class Input {
public:
    typedef size_t size_type;

    explicit Input(FILE* fp, bool owner = false);
    explicit Input(const std::string& path);
    Input(const Input&) = delete;
    Input& operator=(const Input&) = delete;
    ~Input();

    explicit operator bool() const { return m_fp != nullptr; }

    bool Has() const { return !m_eof; }
    std::string Get();
    // Zero-copy form of Get(): pointer/length stay valid until the next call.
    bool GetView(const char** p, size_t* n);
    // Bulk form: the next run of COMPLETE lines (each ending in '\n') as one span, or -- at the
    // end of the stream -- whatever is left, with *final set (the last line then needs no
    // '\n').  After the final span Has() is false.  Lines longer than the buffer grow it.
    bool GetBlock(const char** p, size_t* n, bool* final);

    size_type Size() const { return m_size; }
    size_type Count() const { return m_count; }

private:
    bool Fill();
    size_t ReadSome(char* dst, size_t n);   // fread, or a copy out of the mapping

    FILE* m_fp;
    bool m_owner, m_eof;
    size_type m_size, m_count;
    std::vector<char> m_buf;
    size_t m_pos, m_end;
    std::string m_carry;
    // a regular file on the stream is mapped instead of read (GetBlock then hands out spans of the
    // mapping itself: no copy at all between the page cache and the packer)
    const char* m_map = nullptr;
    size_t m_map_len = 0, m_map_pos = 0;
};

}  // namespace vkmr
