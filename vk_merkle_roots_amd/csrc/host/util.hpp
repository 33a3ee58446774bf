// util.hpp -- small helpers shared by the host code: power-of-two arithmetic
// (reference src/vkmr/Utils.h:22-73), hex printing (print_bytes, src/vkmr/Debug.cpp:38-46)
// and a monotonic stopwatch (src/vkmr/StopWatch.cpp:40-55).
#pragma once
#include <cstdint>
#include <ctime>
#include <string>

namespace vkmr {

template <typename T>
inline bool is_pow2(T v) { return v != 0 && (v & (v - 1)) == 0; }

template <typename T>
inline T largest_pow2_le(T limit)
{
    T r = 1;
    while (limit >= 2 && r <= limit / 2) r *= 2;
    return r;
}

// floor(log2(v)); 0xFFFFFFFF for 0 (reference ln2, src/vkmr/Utils.cpp:11-24).
inline uint32_t ln2(uint64_t v)
{
    if (v == 0) return 0xFFFFFFFFu;
    uint32_t n = 0;
    while (v >>= 1) ++n;
    return n;
}

// Levels of the duplicate-last tree over `count` leaves, at least one (a lone leaf is
// hashed with itself: CpuSha256D::Root's do-while, reference src/vkmr/SHA-256plus.cpp:515-547).
inline uint32_t tree_height(uint64_t count)
{
    uint32_t h = 0;
    while (h < 63 && ((count + ((1ull << h) - 1)) >> h) > 1) ++h;
    return h == 0 ? 1u : h;
}

inline std::string to_hex(const unsigned char* p, size_t n)
{
    static const char d[] = "0123456789abcdef";
    std::string s;
    s.reserve(2 * n);
    for (size_t i = 0; i < n; ++i) {
        s.push_back(d[p[i] >> 4]);
        s.push_back(d[p[i] & 15]);
    }
    return s;
}

class StopWatch {
public:
    bool Start() { return clock_gettime(CLOCK_MONOTONIC, &m_t0) == 0; }
    // milliseconds since Start()
    double Elapsed() const
    {
        timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        return (double)(now.tv_sec - m_t0.tv_sec) * 1e3 + (double)(now.tv_nsec - m_t0.tv_nsec) / 1e6;
    }

private:
    timespec m_t0{};
};

}  // namespace vkmr
