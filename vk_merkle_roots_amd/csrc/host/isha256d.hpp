// isha256d.hpp -- the backend interface of the front end.
//
// Same shape as the reference's vkmr::ISha256D (src/vkmr/ISha256D.h:18-37): a named
// backend that accepts strings one at a time and returns the hex Merkle root.
#pragma once
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

protected:
    name_type m_name;
};

}  // namespace vkmr
