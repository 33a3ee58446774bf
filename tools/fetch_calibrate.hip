// fetch_calibrate.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for the two read patterns of this engine,
// on known byte counts (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte
// count in your own access pattern"):
//   stream   every lane loads 16 contiguous bytes, a wave covers 1 KiB per instruction (the map kernel's staging)
//   pairs    every lane loads 64 contiguous bytes as 4 x 16 B, a wave-instruction touches 16 B of every 64 B over
//            4 KiB (reduce_pass_kernel's pair load, reduce_kernels.hpp: load_node(in + 2j), load_node(in + 2j + 1))
//   pairs_t  the same 4 KiB per wave read as 4 contiguous 1 KiB runs (what a transposed load would issue)
// Each kernel reads `bytes` exactly once (1 GiB by default: beyond the 256 MiB Infinity Cache).
// Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/fetch_calibrate
//             rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- tools/fetch_calibrate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void calib_stream(const uint4* __restrict__ in, uint32_t* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint4 v = in[i];
    const uint32_t x = v.x ^ v.y ^ v.z ^ v.w;
    if (x == 0x12345678u) out[0] = x;   // keeps the load alive without writing
}

__global__ __launch_bounds__(256) void calib_pairs(const uint4* __restrict__ in, uint32_t* __restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;   // 64 B per lane
    const uint4 a = in[i], b = in[i + 1], c = in[i + 2], d = in[i + 3];
    const uint32_t x = a.x ^ b.y ^ c.z ^ d.w ^ a.w ^ b.x ^ c.y ^ d.z;
    if (x == 0x12345678u) out[0] = x;
}

__global__ __launch_bounds__(256) void calib_pairs_t(const uint4* __restrict__ in, uint32_t* __restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint4* p = in + wave * 256 + lane;   // this wave's 4 KiB: 4 runs of 64 x 16 B
    const uint4 a = p[0], b = p[64], c = p[128], d = p[192];
    const uint32_t x = a.x ^ b.y ^ c.z ^ d.w ^ a.w ^ b.x ^ c.y ^ d.z;
    if (x == 0x12345678u) out[0] = x;
}

// The long-string map kernel's walk, without the hashing: lane i streams through its own 2 KiB region.
//   walk64    64 bytes per step from a 4-byte-aligned start (what the per-lane dwordx4 mode does: 4 loads per block)
//   walk128   128-byte lines, line-aligned, one line per step (8 loads): every line is requested exactly once
// WAVES_APART = how far apart in time the steps of one lane are is set by `spin` dependent VALU ops between steps.
template <int BYTES_PER_STEP, bool ALIGNED>
__global__ __launch_bounds__(256) void calib_walk(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t region_words, int spin)
{
    const size_t lane = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t start = lane * region_words + (ALIGNED ? 0 : (lane * 7u) % 32u);   // word index
    typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    uint32_t acc = 0;
    constexpr int WORDS = BYTES_PER_STEP / 4;
    for (uint32_t w = 0; w + WORDS <= region_words - 32; w += WORDS) {
        const u32x4_u* p = reinterpret_cast<const u32x4_u*>(in + start + w);
#pragma unroll
        for (int q = 0; q < WORDS / 4; ++q) {
            const u32x4_u v = p[q];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
        for (int k = 0; k < spin; ++k) acc = acc * 1664525u + 1013904223u;   // dependent chain: the "hash" between steps
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 1024) << 20;
    uint4* in = nullptr;
    uint32_t* out = nullptr;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(in, 0x5a, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_stream, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, 0, in, out);
        hipLaunchKernelGGL(calib_pairs, dim3((unsigned)(bytes / 64 / 256)), dim3(256), 0, 0, in, out);
        hipLaunchKernelGGL(calib_pairs_t, dim3((unsigned)(bytes / 64 / 256)), dim3(256), 0, 0, in, out);
    }
    hipDeviceSynchronize();
    // the walks: 2^21 lanes x 2 KiB = 4 GiB (needs the 4 GiB buffer: run with argument 4096)
    if (bytes >= ((size_t)4 << 30)) {
        const uint32_t region_words = 512;
        const unsigned grid = (unsigned)((bytes / 4 / region_words) / 256);
        for (int spin : {0, 400}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t e[5];
                for (auto& x : e) hipEventCreate(&x);
                hipEventRecord(e[0], 0);
                hipLaunchKernelGGL((calib_walk<64, false>), dim3(grid), dim3(256), 0, 0, reinterpret_cast<const uint32_t*>(in), out, region_words, spin);
                hipEventRecord(e[1], 0);
                hipLaunchKernelGGL((calib_walk<64, true>), dim3(grid), dim3(256), 0, 0, reinterpret_cast<const uint32_t*>(in), out, region_words, spin);
                hipEventRecord(e[2], 0);
                hipLaunchKernelGGL((calib_walk<128, true>), dim3(grid), dim3(256), 0, 0, reinterpret_cast<const uint32_t*>(in), out, region_words, spin);
                hipEventRecord(e[3], 0);
                hipLaunchKernelGGL((calib_walk<128, false>), dim3(grid), dim3(256), 0, 0, reinterpret_cast<const uint32_t*>(in), out, region_words, spin);
                hipEventRecord(e[4], 0);
                hipDeviceSynchronize();
                float a, b, c, d;
                hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]); hipEventElapsedTime(&c, e[2], e[3]); hipEventElapsedTime(&d, e[3], e[4]);
                printf("walk over %u lanes x 2 KiB, spin %d: 64 B unaligned %.3f ms | 64 B aligned %.3f ms | 128 B line-aligned %.3f ms | 128 B unaligned %.3f ms\n",
                       grid * 256, spin, a, b, c, d);
            }
        }
    }
    printf("each kernel read %zu bytes\n", bytes);
    return 0;
}
