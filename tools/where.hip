// where.hip -- which CU does a workgroup land on?  One workgroup = one wavefront; lane 0 stores XCC_ID and HW_ID and the wavefront
// then spins ~20 us so that the launch spreads over every CU its stream may use.  Used by tools/cu_mask_probe.py.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/libwhere.so tools/where.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void where_kernel(uint32_t* out)
{
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 40000ull) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
}

extern "C" int where_launch(hipStream_t s, uint32_t* out, uint32_t nwg)
{
    hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(64), 0, s, out);
    return (int)hipGetLastError();
}
