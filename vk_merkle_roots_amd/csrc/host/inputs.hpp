// inputs.hpp -- line reader for the input stream.
//
// Same contract as the reference's vkmr::Input (src/vkmr/Inputs.h:20-47,
// src/vkmr/Inputs.cpp:52-101): '\n' or end of file ends a line, '\r' is kept, Has() is
// "not at end of file yet" -- so a stream that ends in '\n' yields one final empty
// string, which the caller skips with a warning (src/vkmr/Vkmr.cpp:40-43).  The
// reference reads with one fgetc + append per byte; this reader maps a regular file, and
// takes anything else (the reference's own usage is `rndm ... | vkmr`, README.md:40) in
// blocks of 8 MiB and more with read(2) -- on a thread of its own once the bulk form is in
// use, so that the next block arrives while the caller packs this one.
#pragma once
#include <cstdio>
#include <memory>
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

    // errno of a read that FAILED (not one that reached the end of the stream), else 0.  A stream cut short by an error still
    // ends -- Has() turns false -- and the caller decides: vkmr prints the error and no root (a root over a truncated input
    // would look like an answer).
    int Error() const;

    size_type Size() const { return m_size; }
    size_type Count() const { return m_count; }

private:
    bool Fill();
    size_t ReadSome(char* dst, size_t n);   // read(2) (whatever is there, at least one byte unless the stream has ended), or a copy out of the mapping
    struct Reader;                           // the reading thread of the bulk form and the blocks it shares with the caller
    std::shared_ptr<Reader> m_reader;

    FILE* m_fp;
    int m_error = 0;
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
