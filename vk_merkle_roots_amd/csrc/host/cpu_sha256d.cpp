// cpu_sha256d.cpp -- see cpu_sha256d.hpp.  FIPS 180-4; reference behaviour cited per function.
#include "cpu_sha256d.hpp"

#include <cstring>

#include "util.hpp"

namespace vkmr {
namespace {

const uint32_t kK[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

const uint32_t kIV[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};

inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// One block, 16-word rolling schedule.
void transform(uint32_t st[8], const uint32_t block[16])
{
    uint32_t w[16];
    std::memcpy(w, block, sizeof w);
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int t = 0; t < 64; ++t) {
        if (t >= 16) {
            const uint32_t x = w[(t + 1) & 15], y = w[(t + 14) & 15];
            w[t & 15] += (ror(x, 7) ^ ror(x, 18) ^ (x >> 3)) + w[(t + 9) & 15] + (ror(y, 17) ^ ror(y, 19) ^ (y >> 10));
        }
        const uint32_t t1 = h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + (g ^ (e & (f ^ g))) + kK[t] + w[t & 15];
        const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) | (c & (a | b)));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

}  // namespace

// Reference cpu_sha256_n, src/vkmr/SHA-256plus.cpp:119-276.
void cpu_sha256_words(const unsigned char* msg, size_t len, uint32_t out[8])
{
    uint32_t st[8], blk[16];
    std::memcpy(st, kIV, sizeof st);
    size_t off = 0;
    for (; off + 64 <= len; off += 64) {
        for (int i = 0; i < 16; ++i) blk[i] = be32(msg + off + 4 * i);
        transform(st, blk);
    }
    unsigned char tail[128] = {0};
    const size_t rem = len - off;
    if (rem) std::memcpy(tail, msg + off, rem);
    tail[rem] = 0x80;
    const size_t total = (rem + 9 <= 64) ? 64 : 128;
    const uint64_t bits = (uint64_t)len * 8u;
    for (int i = 0; i < 8; ++i) tail[total - 1 - i] = (unsigned char)(bits >> (8 * i));
    for (size_t o = 0; o < total; o += 64) {
        for (int i = 0; i < 16; ++i) blk[i] = be32(tail + o + 4 * i);
        transform(st, blk);
    }
    std::memcpy(out, st, sizeof st);
}

// Reference cpu_sha256_1 on the result of cpu_sha256_n, src/vkmr/SHA-256plus.cpp:278-358, :479.
void cpu_sha256d_words(const unsigned char* msg, size_t len, uint32_t out[8])
{
    uint32_t first[8], blk[16] = {0};
    cpu_sha256_words(msg, len, first);
    std::memcpy(blk, first, 32);
    blk[8] = 0x80000000u;
    blk[15] = 256u;
    std::memcpy(out, kIV, 32);
    transform(out, blk);
}

// Reference cpu_sha256_2 then cpu_sha256_1, src/vkmr/SHA-256plus.cpp:360-451, :528-530.
void cpu_sha256d_pair(const uint32_t l[8], const uint32_t r[8], uint32_t out[8])
{
    uint32_t st[8], blk[16];
    std::memcpy(st, kIV, 32);
    std::memcpy(blk, l, 32);
    std::memcpy(blk + 8, r, 32);
    transform(st, blk);
    std::memset(blk, 0, sizeof blk);
    blk[0] = 0x80000000u;
    blk[15] = 512u;
    transform(st, blk);
    std::memcpy(blk, st, 32);
    std::memset(blk + 8, 0, 32);
    blk[8] = 0x80000000u;
    blk[15] = 256u;
    std::memcpy(out, kIV, 32);
    transform(out, blk);
}

// Reference CpuSha256D::Root's loop, src/vkmr/SHA-256plus.cpp:513-547.
void cpu_merkle_root_inplace(uint32_t* nodes, size_t n)
{
    do {
        const size_t pairs = (n + 1) / 2;
        for (size_t p = 0; p < pairs; ++p) {
            const uint32_t* l = nodes + 16 * p;
            const uint32_t* r = (2 * p + 1 < n) ? l + 8 : l;
            uint32_t h[8];
            cpu_sha256d_pair(l, r, h);
            std::memcpy(nodes + 8 * p, h, 32);
        }
        n = pairs;
    } while (n > 1);
}

std::string digest_words_to_hex(const uint32_t w[8])
{
    unsigned char bytes[32];
    for (int i = 0; i < 8; ++i) {
        bytes[4 * i] = (unsigned char)(w[i] >> 24);
        bytes[4 * i + 1] = (unsigned char)(w[i] >> 16);
        bytes[4 * i + 2] = (unsigned char)(w[i] >> 8);
        bytes[4 * i + 3] = (unsigned char)w[i];
    }
    return to_hex(bytes, 32);
}

ISha256D::out_type CpuSha256D::Root()
{
    if (m_leaves.empty()) return "";   // reference SHA-256plus.cpp:494-496
    // single-shot like the reference: the leaves are consumed
    cpu_merkle_root_inplace(m_leaves.data(), m_leaves.size() / 8);
    const std::string hex = digest_words_to_hex(m_leaves.data());
    m_leaves.clear();
    return hex;
}

bool CpuSha256D::Add(const arg_type& arg)
{
    uint32_t d[8];
    cpu_sha256d_words(reinterpret_cast<const unsigned char*>(arg.data()), arg.size(), d);
    m_leaves.insert(m_leaves.end(), d, d + 8);
    return true;
}

bool CpuSha256D::AddDigest(const uint32_t words[8])
{
    m_leaves.insert(m_leaves.end(), words, words + 8);
    return true;
}

}  // namespace vkmr
