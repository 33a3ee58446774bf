// isha256d.hpp -- the backend interface of the front end.
//
// Same shape as the reference's vkmr::ISha256D (src/vkmr/ISha256D.h:18-37): a named
// backend that accepts strings one at a time and returns the hex Merkle root.
#pragma once
#include <cstring>
#include <string>

namespace vkmr {

class ISha256D {
public:
    typedef std::string arg_type;
    typedef std::string out_type;
    typedef std::string name_type;

    explicit ISha256D(const name_type& name) : m_name(name) {}
    virtual ~ISha256D() = default;

    const name_type& Name() const { return m_name; }

    // Hex root of everything added so far; "" when nothing was added or on failure.
    virtual out_type Root() = 0;
    // false stops the caller's input loop (reference src/vkmr/Vkmr.cpp:44-47).
    virtual bool Add(const arg_type& arg) = 0;
    // Same as Add() for callers that hold the bytes elsewhere (no temporary string).
    virtual bool Add(const char* bytes, size_t size) { return Add(arg_type(bytes, size)); }
    virtual bool Reset() = 0;

    // Bulk form of the input loop's body: every non-empty line of buf[0,len) is added (a
    // line ends at '\n'; when `final`, the last line may end at len).  Counts what was added
    // and how many empty lines were skipped.  false = an Add was refused; lines before it
    // were added.  The default walks the span line by line.
    struct Tally { size_t items = 0, bytes = 0, empties = 0; };
    virtual bool AddLines(const char* buf, size_t len, bool final, Tally* tally);

protected:
    name_type m_name;
};

inline bool ISha256D::AddLines(const char* buf, size_t len, bool final, Tally* tally)
{
    size_t pos = 0;
    while (pos < len) {
        const char* nl = static_cast<const char*>(memchr(buf + pos, '\n', len - pos));
        if (!nl && !final) break;
        const size_t end = nl ? (size_t)(nl - buf) : len;
        if (end == pos) {
            ++tally->empties;
        } else {
            if (!Add(buf + pos, end - pos)) return false;
            ++tally->items;
            tally->bytes += end - pos;
        }
        pos = nl ? end + 1 : end;
    }
    return true;
}

}  // namespace vkmr
