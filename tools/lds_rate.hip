// lds_rate.hip -- LDS operation rates on gfx950 (per-lane private cells, conflict-free layout):
// ds_read_b32, ds_write_b32, ds_add_u32 (no return), ds_add_rtn_u32.  Used to decide whether part of
// the SHA-256 message-schedule additions could be moved to the LDS atomic ALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters)
{
    __shared__ uint32_t cell[32 * 256];
    uint32_t* mine = cell + threadIdx.x;
    for (int i = 0; i < 32; ++i) mine[i * 256] = threadIdx.x + i;
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (KIND == 0) acc += mine[i * 256];
            if (KIND == 1) mine[i * 256] = acc + i;
            if (KIND == 2) __hip_atomic_fetch_add(&mine[i * 256], acc | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (KIND == 3) acc += __hip_atomic_fetch_add(&mine[i * 256], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (KIND == 0 || KIND == 3) acc = acc * 3u + 1u; else asm volatile("" ::: "memory");
    }
    uint32_t s = acc;
    for (int i = 0; i < 32; ++i) s ^= mine[i * 256];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, uint32_t* d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double waveops = (double)blocks * 4 * iters * 32;
    printf("%-18s %8.3f ms  %7.2f G wave-ops/s chip  = %.2f wave-ops per CU per 100 clk @2.1GHz\n", name, ms, waveops / (ms * 1e-3) / 1e9,
           waveops / (ms * 1e-3) / 256 / 2.1e9 * 100);
}

int main()
{
    uint32_t* d;
    (void)hipMalloc(&d, 256 * 4096 * 4);
    for (int bpc : {1, 4}) {
        printf("-- %d workgroups of 256 per CU\n", bpc);
        run<0>("ds_read_b32", d, 256 * bpc, 2000);
        run<1>("ds_write_b32", d, 256 * bpc, 2000);
        run<2>("ds_add_u32", d, 256 * bpc, 2000);
        run<3>("ds_add_rtn_u32", d, 256 * bpc, 2000);
    }
    return 0;
}
