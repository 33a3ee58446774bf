// stamps.hpp -- DIAGNOSTIC BUILD ONLY (-DVKMR_STAMPS, vk_merkle_roots_amd/libvkmr_hip_stamps.so; tools/kernel_clock.py).
// Lane 0 of every workgroup's first wavefront records shader-clock (s_memtime) and constant 100 MHz
// (s_memrealtime) stamps at its phase boundaries into a buffer of their own: 8 words per workgroup, no atomics,
// read by no kernel.  The in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup
// (MI355X_MICROARCH.md, "DVFS give-back" item 6).  In the product build no stamp executes.
#pragma once
#ifdef VKMR_STAMPS
#define VKMR_STAMP_SLOTS 65536
__device__ unsigned long long g_stamps[VKMR_STAMP_SLOTS * 8];
#define VKMR_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#define VKMR_STAMP_RT(var) unsigned long long var = __builtin_amdgcn_s_memrealtime()
#else
#define VKMR_STAMP(var)
#define VKMR_STAMP_RT(var)
#endif
