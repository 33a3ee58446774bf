// host_api.cpp -- C entry points over the host-side C++ for bindings and tests
// (libvkmr_host.so): the "CPU" backend on packed batches and on slice roots.
#include <cstdint>
#include <cstring>
#include <vector>

#include "cpu_sha256d.hpp"
#include "vkmr_hip.h"

extern "C" {

// Leaf digests of a packed batch with the CPU backend (CpuSha256D::Add per string).
__attribute__((visibility("default"))) void vkmr_host_cpu_leaves(const uint32_t* data, const vkmr_metadata* meta, uint64_t count,
                                                                  vkmr_digest* out)
{
    for (uint64_t i = 0; i < count; ++i)
        vkmr::cpu_sha256d_words(reinterpret_cast<const unsigned char*>(data + meta[i].start), meta[i].size, out[i].data);
}

// Sub-tree root of `count` digests through exactly `height` levels (the per-slice
// contract of vkmr_hip_reduce_async), on the CPU backend's node function.
__attribute__((visibility("default"))) int vkmr_host_cpu_reduce(const vkmr_digest* digests, uint64_t count, uint32_t height,
                                                                 vkmr_digest* root)
{
    if (!digests || !root || count == 0) return -1;
    std::vector<uint32_t> nodes(8 * count);
    std::memcpy(nodes.data(), digests, 32 * count);
    uint64_t n = count;
    for (uint32_t lv = 0; lv < height; ++lv) {
        const uint64_t pairs = (n + 1) / 2;
        for (uint64_t p = 0; p < pairs; ++p) {
            const uint32_t* l = nodes.data() + 16 * p;
            const uint32_t* r = (2 * p + 1 < n) ? l + 8 : l;
            uint32_t h[8];
            vkmr::cpu_sha256d_pair(l, r, h);
            std::memcpy(nodes.data() + 8 * p, h, 32);
        }
        n = pairs;
    }
    if (n != 1) return -1;
    std::memcpy(root->data, nodes.data(), 32);
    return 0;
}

// Combine of slice roots in slice order: duplicate-last tree, at least one level
// (CpuSha256D::Root over already-hashed nodes; reference CpuSha256DforReductions,
// src/vkmr/Reductions.cpp:56-69, :703-712).
__attribute__((visibility("default"))) int vkmr_host_cpu_combine(const vkmr_digest* roots, uint32_t n, vkmr_digest* out)
{
    if (!roots || !out || n == 0) return -1;
    std::vector<uint32_t> nodes(8 * (size_t)n);
    std::memcpy(nodes.data(), roots, 32 * (size_t)n);
    vkmr::cpu_merkle_root_inplace(nodes.data(), n);
    std::memcpy(out->data, nodes.data(), 32);
    return 0;
}

// Folds a leaf digest with the siblings of its authentication path (vkmr_hip_proof_async):
// bit l of `index` tells whether the path node is the right (1) or left (0) operand at level l.
__attribute__((visibility("default"))) void vkmr_host_cpu_fold_proof(const vkmr_digest* leaf, uint64_t index, const vkmr_digest* siblings,
                                                                      uint32_t height, vkmr_digest* root)
{
    uint32_t cur[8];
    std::memcpy(cur, leaf->data, 32);
    for (uint32_t l = 0; l < height; ++l) {
        uint32_t next[8];
        const bool right = (l < 64) && ((index >> l) & 1ull);
        if (right)
            vkmr::cpu_sha256d_pair(siblings[l].data, cur, next);
        else
            vkmr::cpu_sha256d_pair(cur, siblings[l].data, next);
        std::memcpy(cur, next, 32);
    }
    std::memcpy(root->data, cur, 32);
}

}  // extern "C"
