// sha256d_lds.hpp -- EXPERIMENTS BUILD ONLY (-DVKMR_EXPERIMENTS; tools/sha_variants.hip and the SCHED variants of
// map_kernel).  The north star's form of the compression: round constants K[64] read from LDS (SCHED 1), and the
// 16-word message-schedule ring kept in LDS as well (SCHED 2), against the shipped form (K as literals/SGPRs, ring
// in VGPRs with static indices; sha256d_device.hpp).  Reference form: src/shaders/SHA-256.comp:112-152 (M[16], W[64]
// and c_constants[64] per invocation).  Measured: profiles/r01_lds_schedule_constants_ab.txt (node hash),
// profiles/r03_map_lds_schedule_ab.txt (inside map_kernel).
#pragma once
#include "../sha256d_device.hpp"

namespace vkmr_dev {

// sW: this lane's ring cell j lives at sW[j * STRIDE] (STRIDE = lanes per workgroup: conflict-free)
template <int T, bool W_LDS, int STRIDE>
__device__ __forceinline__ void lds_round(uint32_t (&s)[8], uint32_t (&w)[16], const uint32_t* sK, uint32_t* sW)
{
    constexpr int i = T & 15;
    uint32_t wt;
    if (W_LDS) {
        if (T >= 16) {
            const uint32_t w0 = sW[i * STRIDE], w1 = sW[((i + 1) & 15) * STRIDE], w9 = sW[((i + 9) & 15) * STRIDE], w14 = sW[((i + 14) & 15) * STRIDE];
            wt = w0 + ssig0(w1) + w9 + ssig1(w14);
            sW[i * STRIDE] = wt;
        } else {
            wt = sW[i * STRIDE];
        }
    } else {
        if (T >= 16) w[i] = w[i] + ssig0(w[(i + 1) & 15]) + w[(i + 9) & 15] + ssig1(w[(i + 14) & 15]);
        wt = w[i];
    }
    round_fn<T>(s, sK[T] + wt);
}

template <bool W_LDS, int STRIDE, int... T>
__device__ __forceinline__ void lds_rounds(uint32_t (&s)[8], uint32_t (&w)[16], const uint32_t* sK, uint32_t* sW, std::integer_sequence<int, T...>)
{
    (lds_round<T, W_LDS, STRIDE>(s, w, sK, sW), ...);
}

template <bool W_LDS, int STRIDE>
__device__ __forceinline__ void lds_compress(uint32_t (&H)[8], uint32_t (&w)[16], const uint32_t* sK, uint32_t* sW)
{
    uint32_t s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = H[i];
    if (W_LDS) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sW[i * STRIDE] = w[i];
    }
    lds_rounds<W_LDS, STRIDE>(s, w, sK, sW, std::make_integer_sequence<int, 64>{});
#pragma unroll
    for (int i = 0; i < 8; ++i) H[i] += s[i];
}

template <bool W_LDS, int STRIDE>
__device__ __forceinline__ void lds_hash_digest(const uint32_t (&in)[8], uint32_t (&out)[8], const uint32_t* sK, uint32_t* sW)
{
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = in[i];
    w[8] = 0x80000000u;
#pragma unroll
    for (int i = 9; i < 15; ++i) w[i] = 0u;
    w[15] = 256u;
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = IV256[i];
    lds_compress<W_LDS, STRIDE>(out, w, sK, sW);
}

}  // namespace vkmr_dev
