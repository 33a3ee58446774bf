// rndm_stream.hpp -- the `rndm` input generator as a library.
//
// The reference's rndm tool (src/rndm/Rndm.cpp:20-71) draws everything from libc
// rand() after srand(seed): per string `len = 1 + rand() % (max - 1)`, then per
// byte `32 + rand() % 94`.  On glibc that is the TYPE_3 additive-feedback
// generator; it is restated here (own state, no lock, ~1 ns per draw) so that a
// 2^26-string stream can be produced in seconds, straight into packed batches,
// and stays bit-identical to `rndm <seed> <count> <max>` piped through stdin.
// tests/test_host_tools.py checks the draws against libc rand() itself.
#pragma once
#include <cstdint>

namespace vkmr {

class GlibcRand {
public:
    explicit GlibcRand(uint32_t seed);
    // Next value of rand(): 31 bits.
    inline uint32_t Next()
    {
        uint32_t v = m_r[m_front] + m_r[m_rear];
        m_r[m_front] = v;
        m_front = (m_front + 1 == 31) ? 0 : m_front + 1;
        m_rear = (m_rear + 1 == 31) ? 0 : m_rear + 1;
        return v >> 1;
    }

private:
    uint32_t m_r[31];
    int m_front, m_rear;
};

}  // namespace vkmr
