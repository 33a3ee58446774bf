// cpu_sha256d.hpp -- the "CPU" backend: serial SHA-256d Merkle root on the host.
//
// Drop-in for the reference's vkmr::CpuSha256D (src/vkmr/SHA-256plus.h:32-49): same
// name ("CPU"), same Add/Root/Reset contract, same tree (duplicate-last, a lone leaf
// hashed with itself).  This is the product's own serial backend (independent of the test
// checker), written around one streaming compression function; nodes are kept
// as 8 words in a flat vector instead of a vector per node.
#pragma once
#include <cstdint>
#include <vector>

#include "isha256d.hpp"

namespace vkmr {

// SHA-256 of `len` bytes as word values H[0..7].
void cpu_sha256_words(const unsigned char* msg, size_t len, uint32_t out[8]);
// Leaf digest SHA-256(SHA-256(bytes)) as word values (reference cpu_sha256d_int, SHA-256plus.cpp:479).
void cpu_sha256d_words(const unsigned char* msg, size_t len, uint32_t out[8]);
// Node SHA-256d(l || r) on word values (reference cpu_sha256_1(cpu_sha256_2(l, r)), SHA-256plus.cpp:528-530).
void cpu_sha256d_pair(const uint32_t l[8], const uint32_t r[8], uint32_t out[8]);
// Duplicate-last root over n >= 1 nodes (8 words each), in place; result in nodes[0..7].
void cpu_merkle_root_inplace(uint32_t* nodes, size_t n);
// Canonical hex of a word-valued digest (hash_to_string + print_bytes, SHA-256plus.cpp:453-469, :555).
std::string digest_words_to_hex(const uint32_t w[8]);

class CpuSha256D : public ISha256D {
public:
    CpuSha256D() : ISha256D("CPU") {}

    out_type Root() override;
    bool Add(const arg_type& arg) override;
    // Feeds an already-hashed node (slice roots from the GPU; reference
    // CpuSha256DforReductions::Add, src/vkmr/Reductions.cpp:56-69).
    bool AddDigest(const uint32_t words[8]);
    bool Reset() override
    {
        m_leaves.clear();
        return true;
    }

private:
    std::vector<uint32_t> m_leaves;   // 8 words per leaf
};

}  // namespace vkmr
