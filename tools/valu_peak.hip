// valu_peak.hip -- measures the sustained issue rate of the int32 VALU instructions the
// SHA-256 round is made of (v_alignbit_b32, v_bitop3_b32, v_add3_u32, v_add_u32) on the
// whole chip, to price the kernels against what the hardware can actually issue.
//   hipcc -O3 --offload-arch=gfx950 -o valu_peak tools/valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(256) void spin(uint32_t* out, int iters)
{
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 1) & 7], 7);
                if (KIND == 1) a[i] = __builtin_amdgcn_bitop3_b32(a[i], a[(i + 1) & 7], a[(i + 2) & 7], 0x96);
                if (KIND == 2) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 3) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 4) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 5) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 6) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 7) asm volatile("v_lshl_or_b32 %0, %1, 7, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 8) asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 9) asm volatile("v_lshrrev_b32 %0, 7, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 10) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 11) asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 12) asm volatile("v_bfe_u32 %0, %1, 3, 9" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 14) asm volatile("v_alignbyte_b32 %0, %1, %2, 1" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 15) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 16) asm volatile("v_lshrrev_b64 %0, 7, %1" : "=v"(*(unsigned long long*)&a[i & 6]) : "v"(*(unsigned long long*)&a[(i + 2) & 6]));
                if (KIND == 17) asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 19) asm volatile("v_alignbit_b32 %0, %1, %1, 7" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 20) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "s"(iters));
                if (KIND == 21) asm volatile("v_add3_u32 %0, %1, %1, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 22) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 24) asm volatile("v_lshlrev_b32 %0, 7, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 25) asm volatile("v_alignbit_b32 %0, %1, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(iters));
                if (KIND == 26) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 27) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 28) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(unsigned long long*)&a[i & 6]) : "v"(*(unsigned long long*)&a[(i + 2) & 6]), "v"(*(unsigned long long*)&a[(i + 4) & 6]));
                if (KIND == 29) asm volatile("v_add_co_u32 %0, vcc, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
                if (KIND == 30) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 31) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(unsigned long long*)&a[i & 6]) : "v"(*(unsigned long long*)&a[(i + 2) & 6]), "v"(*(unsigned long long*)&a[(i + 4) & 6]));
                if (KIND == 32) asm volatile("v_cmp_lt_u32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
                if (KIND == 33) asm volatile("v_cmp_lt_u32_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]) : "s20", "s21");
                if (KIND == 34) asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 35) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 36) asm volatile("v_ashrrev_i32 %0, 7, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 37) asm volatile("v_min_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 38) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
                if (KIND == 39) asm volatile("v_or_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                if (KIND == 40) asm volatile("v_add_u32 %0, 0x428a2f98, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 41) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "s"(iters));
                if (KIND == 18) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
double run(const char* name, int blocks, int iters, uint32_t* d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 128.0;
    double tops = ops / (ms * 1e-3) / 1e12;
    printf("%-16s blocks=%5d iters=%d  %.3f ms  %.2f T lane-ops/s\n", name, blocks, iters, ms, tops);
    return tops;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device: %s arch=%s CUs=%d clock=%d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    uint32_t* d;
    hipMalloc(&d, 256 * 8192 * 4);
    for (int wavesPerSimd : {4}) {
        int blocks = p.multiProcessorCount * wavesPerSimd;
        printf("-- %d wave(s) per SIMD, more instruction kinds\n", wavesPerSimd);
        run<6>("v_perm_b32", blocks, 4000, d);
        run<7>("v_lshl_or_b32", blocks, 4000, d);
        run<8>("v_xad_u32", blocks, 4000, d);
        run<9>("v_lshrrev_b32", blocks, 4000, d);
        run<10>("v_and_or_b32", blocks, 4000, d);
        run<11>("v_lshl_add_u32", blocks, 4000, d);
        run<12>("v_bfe_u32", blocks, 4000, d);
        run<14>("v_alignbyte_b32", blocks, 4000, d);
        run<15>("v_pk_add_u16", blocks, 4000, d);
        run<16>("v_lshrrev_b64", blocks, 4000, d);
        run<17>("v_mov_b32_dpp", blocks, 4000, d);
        run<18>("v_add_u32_sdwa", blocks, 4000, d);
        run<19>("alignbit same-reg", blocks, 4000, d);
        run<20>("add3 v,v,s", blocks, 4000, d);
        run<21>("add3 same-reg", blocks, 4000, d);
        run<22>("v_bfi_b32", blocks, 4000, d);
        run<24>("v_lshlrev_b32", blocks, 4000, d);
        run<25>("alignbit v,v,s", blocks, 4000, d);
        run<26>("v_mad_u32_u24", blocks, 4000, d);
        run<27>("v_mul_lo_u32", blocks, 4000, d);
        run<28>("v_pk_mul_f32", blocks, 4000, d);
        run<29>("v_add_co_u32", blocks, 4000, d);
        run<30>("v_mov_b32", blocks, 4000, d);
        run<31>("v_pk_add_f32", blocks, 4000, d);
        run<32>("cmp_e32+cndmask_e32 (2 instr)", blocks, 4000, d);
        run<33>("cmp_e64+cndmask_e64 (2 instr)", blocks, 4000, d);
        run<34>("v_and_b32", blocks, 4000, d);
        run<35>("v_sub_u32", blocks, 4000, d);
        run<36>("v_ashrrev_i32", blocks, 4000, d);
        run<37>("v_min_u32", blocks, 4000, d);
        run<38>("v_cmp_lt_u32_e32", blocks, 4000, d);
        run<39>("v_or_b32", blocks, 4000, d);
        run<40>("v_add_u32 literal", blocks, 4000, d);
        run<41>("v_bitop3 v,v,s", blocks, 4000, d);
    }
    for (int wavesPerSimd : {1, 2, 4, 8}) {
        int blocks = p.multiProcessorCount * wavesPerSimd;   // 256-thread blocks = 4 waves = 1 per SIMD
        printf("-- %d wave(s) per SIMD\n", wavesPerSimd);
        run<0>("v_alignbit_b32", blocks, 4000, d);
        run<1>("v_bitop3_b32", blocks, 4000, d);
        run<2>("v_add3_u32", blocks, 4000, d);
        run<3>("v_add_u32", blocks, 4000, d);
        run<5>("v_xor_b32", blocks, 4000, d);
        run<4>("v_fma_f32", blocks, 4000, d);
    }
    return 0;
}
