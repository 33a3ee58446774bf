// sha_variants.hip -- A/B for the north star's "message schedule and round constants staged in
// LDS": node hashes/s of (A) the shipped code (K as literals/SGPRs, 16-word schedule ring in
// VGPRs), (B) K[64] read from LDS, (C) K from LDS and the schedule ring in LDS as well.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/sha_variants tools/sha_variants.hip && ./tools/sha_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../vk_merkle_roots_amd/csrc/experiments/sha256d_lds.hpp"

using namespace vkmr_dev;

template <int VARIANT>
__global__ __launch_bounds__(256) void node_kernel(const Node* in, Node* out, int reps)
{
    __shared__ uint32_t sK[64];
    __shared__ uint32_t sWall[VARIANT == 2 ? 16 * 256 : 1];
    if (threadIdx.x < 64) sK[threadIdx.x] = K256[threadIdx.x];
    __syncthreads();
    uint32_t* sW = sWall + (VARIANT == 2 ? threadIdx.x : 0);
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Node a = load_node(in + 2 * p), b = load_node(in + 2 * p + 1);
    uint32_t l[8], r[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { l[i] = a.w[i]; r[i] = b.w[i]; }
    for (int it = 0; it < reps; ++it) {
        if (VARIANT == 0) {
            hash_pair(l, r, o);
        } else {
            uint32_t w[16], H[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { w[i] = l[i]; w[8 + i] = r[i]; H[i] = IV256[i]; }
            lds_compress<VARIANT == 2, 256>(H, w, sK, sW);
            w[0] = 0x80000000u;
#pragma unroll
            for (int i = 1; i < 15; ++i) w[i] = 0u;
            w[15] = 512u;
            lds_compress<VARIANT == 2, 256>(H, w, sK, sW);
#pragma unroll
            for (int i = 0; i < 8; ++i) { w[i] = H[i]; o[i] = IV256[i]; }
            w[8] = 0x80000000u;
#pragma unroll
            for (int i = 9; i < 15; ++i) w[i] = 0u;
            w[15] = 256u;
            lds_compress<VARIANT == 2, 256>(o, w, sK, sW);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { l[i] = o[i]; r[i] ^= o[i]; }
    }
    store_node(out + p, o);
}

template <int VARIANT>
void run(const char* name, const Node* in, Node* out, int blocks, int reps, uint32_t* check)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(node_kernel<VARIANT>, dim3(blocks), dim3(256), 0, 0, in, out, reps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(node_kernel<VARIANT>, dim3(blocks), dim3(256), 0, 0, in, out, reps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[8];
    (void)hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    const double hashes = (double)blocks * 256 * reps;
    printf("%-44s %8.3f ms  %7.2f G node hashes/s   out[0]=%08x%s\n", name, ms, hashes / (ms * 1e-3) / 1e9, h[0],
           (check[0] && check[0] != h[0]) ? "  MISMATCH" : "");
    if (!check[0]) check[0] = h[0];
}

int main(int argc, char** argv)
{
    // argv[1]: workgroups (4 wavefronts each) per CU -- occupancy sweep: does the chip clock higher with fewer waves?
    const int bpc = argc > 1 ? atoi(argv[1]) : 8;
    const int blocks = 256 * bpc, reps = 64 * 8 / bpc;
    printf("-- %d workgroups of 4 wavefronts per CU (%d waves/SIMD)\n", bpc, bpc);
    Node *in, *out;
    (void)hipMalloc(&in, (size_t)blocks * 256 * 2 * sizeof(Node));
    (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(Node));
    (void)hipMemset(in, 0x5a, (size_t)blocks * 256 * 2 * sizeof(Node));
    uint32_t check[1] = {0};
    for (int round = 0; round < 2; ++round) {
        run<0>("A: K literals/SGPRs, W ring in VGPRs (shipped)", in, out, blocks, reps, check);
        run<1>("B: K[64] in LDS, W ring in VGPRs", in, out, blocks, reps, check);
        run<2>("C: K[64] in LDS, W ring in LDS", in, out, blocks, reps, check);
    }
    return 0;
}
