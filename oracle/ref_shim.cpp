// ref_shim.cpp -- C entry points over the UNMODIFIED reference CPU-serial backend,
// so tests can compare the restatement (sha256d_oracle.c) leaf by leaf and root by
// root with the reference itself.  TEST INFRASTRUCTURE ONLY; built into
// oracle/_ref/libvkmr_ref.so by oracle/Makefile from the reference sources where
// they lie (src/vkmr/SHA-256plus.cpp, Debug.cpp).  Nothing from the reference is
// copied into this repository.
#include <cstdint>
#include <cstring>
#include <string>

#include "SHA-256plus.h"

extern "C" {

// vkmr::cpu_sha256 (src/vkmr/SHA-256plus.cpp:473-477): canonical 32 bytes.
__attribute__((visibility("default"))) void ref_sha256(const uint8_t* msg, size_t len, uint8_t out[32])
{
    const std::string r = vkmr::cpu_sha256(std::string(reinterpret_cast<const char*>(msg), len));
    std::memcpy(out, r.data(), 32);
}

// vkmr::cpu_sha256d (src/vkmr/SHA-256plus.cpp:481-487): canonical 32 bytes.
__attribute__((visibility("default"))) void ref_sha256d(const uint8_t* msg, size_t len, uint8_t out[32])
{
    const std::string r = vkmr::cpu_sha256d(std::string(reinterpret_cast<const char*>(msg), len));
    std::memcpy(out, r.data(), 32);
}

// CpuSha256D::Add per string then Root() (src/vkmr/SHA-256plus.cpp:491-561).
// Strings are given packed: offsets[i]..offsets[i+1] into bytes.  hex gets the
// 64-char root ("" when n == 0).
__attribute__((visibility("default"))) void ref_root(const uint8_t* bytes, const uint64_t* offsets, size_t n, char hex[65])
{
    vkmr::CpuSha256D tree;
    for (size_t i = 0; i < n; ++i)
        tree.Add(std::string(reinterpret_cast<const char*>(bytes + offsets[i]), offsets[i + 1] - offsets[i]));
    const std::string r = tree.Root();
    std::memset(hex, 0, 65);
    std::memcpy(hex, r.data(), r.size() < 64 ? r.size() : 64);
}

}
