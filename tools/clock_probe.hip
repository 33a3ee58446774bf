// clock_probe.hip -- what shader clock does the chip HOLD under VALU-dense integer work, and how much of the
// issue rate at THAT clock do the instruction kinds SHA-256 is made of reach?
//
// Every wavefront stamps s_memtime (shader cycles) and s_memrealtime (constant 100 MHz) around its loop; the
// in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz per wavefront (MI355X_MICROARCH.md, "DVFS
// give-back" item 6: stamped once around the loop after >= 2 s of back-to-back launches).  s_memtime counters
// are per XCD and not synchronised, so only per-wavefront differences are used.  Stamps go to a buffer of their
// own; the measured value never feeds the loop.  Board power and the driver's sclk are read from sysfs hwmon
// beside it (the guide: not the test, but they say whether the board sits at its power cap).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/clock_probe tools/clock_probe.hip && ./tools/clock_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>

#include "../vk_merkle_roots_amd/csrc/sha256d_device.hpp"

enum { K_XOR = 0, K_BITOP3, K_ALIGNBIT, K_ADD3, K_ADD, K_NODE, K_MIX, K_XOR_D1, K_XOR_D2, K_XOR_D4, K_ALIGN_D1, K_ALIGN_D2, K_BITOP_SAMEBANK, K_BITOP_DIFFBANK,
       K_ADD3_SAMEBANK, K_ADD3_DIFFBANK, K_NODE2, K_ROUNDS_DEP, K_ROUNDS_SPACED, K_MIX_GROUPED, K_XOR_E64 };

// Two independent tree-node hashes per lane, interleaved statement by statement (ILP 2): does the issue rate of the
// real code suffer from its tight dependency chains?
namespace ilp2 {
using namespace vkmr_dev;
template <int T>
__device__ __forceinline__ void round2(uint32_t (&s)[2][8], const uint32_t (&kw)[2])
{
    uint32_t t1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) t1[j] = s[j][(7 - T) & 7] + kw[j] + bsig1(s[j][(4 - T) & 7]) + ch(s[j][(4 - T) & 7], s[j][(5 - T) & 7], s[j][(6 - T) & 7]);
#pragma unroll
    for (int j = 0; j < 2; ++j) s[j][(3 - T) & 7] += t1[j];
#pragma unroll
    for (int j = 0; j < 2; ++j) s[j][(7 - T) & 7] = t1[j] + bsig0(s[j][(0 - T) & 7]) + maj(s[j][(0 - T) & 7], s[j][(1 - T) & 7], s[j][(2 - T) & 7]);
}
template <int T>
__device__ __forceinline__ void sched_round2(uint32_t (&s)[2][8], uint32_t (&w)[2][16])
{
    constexpr int i = T & 15;
    uint32_t kw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (T >= 16) w[j][i] = w[j][i] + ssig0(w[j][(i + 1) & 15]) + w[j][(i + 9) & 15] + ssig1(w[j][(i + 14) & 15]);
        kw[j] = K256[T] + w[j][i];
    }
    round2<T>(s, kw);
}
template <int... T>
__device__ __forceinline__ void all_rounds2(uint32_t (&s)[2][8], uint32_t (&w)[2][16], std::integer_sequence<int, T...>)
{
    (sched_round2<T>(s, w), ...);
}
__device__ __forceinline__ void compress2(uint32_t (&H)[2][8], uint32_t (&w)[2][16])
{
    uint32_t s[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) s[j][i] = H[j][i];
    all_rounds2(s, w, std::make_integer_sequence<int, 64>{});
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) H[j][i] += s[j][i];
}
__device__ __forceinline__ void hash_pair2(const uint32_t (&l)[2][8], const uint32_t (&r)[2][8], uint32_t (&out)[2][8])
{
    uint32_t w[2][16], H[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[j][i] = l[j][i]; w[j][8 + i] = r[j][i]; H[j][i] = IV256[i]; }
    compress2(H, w);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        w[j][0] = 0x80000000u;
#pragma unroll
        for (int i = 1; i < 15; ++i) w[j][i] = 0u;
        w[j][15] = 512u;
    }
    compress2(H, w);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[j][i] = H[j][i]; out[j][i] = IV256[i]; }
        w[j][8] = 0x80000000u;
#pragma unroll
        for (int i = 9; i < 15; ++i) w[j][i] = 0u;
        w[j][15] = 256u;
    }
    compress2(out, w);
}
}  // namespace ilp2

// One SHA-256 round's worth of VALU work as a fixed instruction sequence (asm volatile: the compiler cannot reorder).
// DEP: the order hipcc emits for the real code (each consumer right behind its producers); SPACED: the same
// instructions with every consumer at least three instructions behind its producer (two rounds' work interleaved).
#define RND_DEP(e, a, f, g, b, c, h, d, k)                                                      \
    asm volatile(                                                                                  \
        "v_alignbit_b32 %[t0], %[E], %[E], 25\n\tv_alignbit_b32 %[t1], %[E], %[E], 11\n\tv_alignbit_b32 %[t2], %[E], %[E], 6\n\t" \
        "v_bitop3_b32 %[t0], %[t2], %[t1], %[t0] bitop3:0x96\n\tv_bitop3_b32 %[t3], %[E], %[F], %[G] bitop3:0xca\n\t"    \
        "v_add3_u32 %[H], %[H], %[t0], %[t3]\n\t"                                                \
        "v_alignbit_b32 %[t0], %[A], %[A], 22\n\tv_alignbit_b32 %[t1], %[A], %[A], 13\n\tv_alignbit_b32 %[t2], %[A], %[A], 2\n\t" \
        "v_add_u32 %[H], %[K], %[H]\n\t"                                                         \
        "v_bitop3_b32 %[t3], %[A], %[B], %[C] bitop3:0xe8\n\tv_bitop3_b32 %[t0], %[t2], %[t1], %[t0] bitop3:0x96\n\t"    \
        "v_add_u32 %[D], %[H], %[D]\n\tv_add3_u32 %[H], %[t0], %[H], %[t3]\n\t"                  \
        : [H] "+v"(h), [D] "+v"(d), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)  \
        : [E] "v"(e), [F] "v"(f), [G] "v"(g), [A] "v"(a), [B] "v"(b), [C] "v"(c), [K] "v"(k))

template <int KIND>
__global__ __launch_bounds__(256) void spin(uint32_t* out, unsigned long long* stamps, int iters, uint32_t data_mask)
{
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x * 977u) & data_mask;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    if (KIND == K_ROUNDS_DEP) {
        uint32_t t0, t1, t2, t3, k = data_mask | 0x428a2f98u;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                RND_DEP(a[4], a[0], a[5], a[6], a[1], a[2], a[7], a[3], k);
                RND_DEP(a[3], a[7], a[4], a[5], a[0], a[1], a[6], a[2], k);
                RND_DEP(a[2], a[6], a[3], a[4], a[7], a[0], a[5], a[1], k);
                RND_DEP(a[1], a[5], a[2], a[3], a[6], a[7], a[4], a[0], k);
                RND_DEP(a[0], a[4], a[1], a[2], a[5], a[6], a[3], a[7], k);
                RND_DEP(a[7], a[3], a[0], a[1], a[4], a[5], a[2], a[6], k);
                RND_DEP(a[6], a[2], a[7], a[0], a[3], a[4], a[1], a[5], k);
                RND_DEP(a[5], a[1], a[6], a[7], a[2], a[3], a[0], a[4], k);
            }
        }
    } else if (KIND == K_NODE2) {
        uint32_t l[2][8], r[2][8], o[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { l[0][i] = a[i]; r[0][i] = ~a[i] & data_mask; l[1][i] = a[i] * 3u; r[1][i] = a[i] + 77u; }
        for (int it = 0; it < iters; ++it) {
            ilp2::hash_pair2(l, r, o);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) { l[j][i] = o[j][i] & data_mask; r[j][i] ^= o[j][i] & data_mask; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = l[0][i] ^ r[0][i] ^ l[1][i] ^ r[1][i];
    } else if (KIND == K_NODE) {
        // the register-resident tree-node hash (3 compressions), chained
        uint32_t l[8], r[8], o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { l[i] = a[i]; r[i] = ~a[i] & data_mask; }
        for (int it = 0; it < iters; ++it) {
            vkmr_dev::hash_pair(l, r, o);
#pragma unroll
            for (int i = 0; i < 8; ++i) { l[i] = o[i] & data_mask; r[i] ^= o[i] & data_mask; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = l[i] ^ r[i];
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (KIND == K_XOR) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                    if (KIND == K_BITOP3) a[i] = __builtin_amdgcn_bitop3_b32(a[i], a[(i + 1) & 7], a[(i + 2) & 7], 0x96);
                    if (KIND == K_ALIGNBIT) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 1) & 7], 7);
                    if (KIND == K_ADD3) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                    if (KIND == K_ADD) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                    if (KIND == K_XOR_E64) asm volatile("v_xor_b32_e64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                    if (KIND == K_XOR_D1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[0]) : "v"(a[1]));
                    if (KIND == K_XOR_D2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[(i & 1) * 2]) : "v"(a[1]));
                    if (KIND == K_XOR_D4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[(i & 3)]) : "v"(a[7]));
                    if (KIND == K_ALIGN_D1) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(a[0]));
                    if (KIND == K_ALIGN_D2) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(a[i & 1]));
                    if (KIND == K_BITOP_SAMEBANK) asm volatile("v_bitop3_b32 v[40+%c0], v44, v48, v52 bitop3:0x96" : : "n"((i & 3) * 4 ) : "v40", "v44", "v48", "v52");
                    if (KIND == K_BITOP_DIFFBANK) asm volatile("v_bitop3_b32 v[40+%c0], v45, v50, v55 bitop3:0x96" : : "n"((i & 3) * 4) : "v40", "v44", "v48", "v52");
                    if (KIND == K_ADD3_SAMEBANK) asm volatile("v_add3_u32 v[40+%c0], v44, v48, v52" : : "n"((i & 3) * 4) : "v40", "v44", "v48", "v52");
                    if (KIND == K_ADD3_DIFFBANK) asm volatile("v_add3_u32 v[40+%c0], v45, v50, v55" : : "n"((i & 3) * 4) : "v40", "v44", "v48", "v52");
                    if (KIND == K_MIX_GROUPED) {
                        // the mix's 8 instructions with the three rotates of ONE value followed at once by their xor (the real code's shape)
                        if (i == 0) { uint32_t t0, t1, t2; asm volatile("v_alignbit_b32 %0, %3, %3, 25\n\tv_alignbit_b32 %1, %3, %3, 11\n\tv_alignbit_b32 %2, %3, %3, 6\n\tv_bitop3_b32 %0, %2, %1, %0 bitop3:0x96\n\tv_bitop3_b32 %1, %3, %4, %5 bitop3:0xca\n\tv_add3_u32 %3, %3, %0, %1\n\tv_add_u32 %4, %3, %4\n\tv_lshrrev_b32 %5, 3, %4" : "=&v"(t0), "=&v"(t1), "=&v"(t2), "+v"(a[r & 7]), "+v"(a[(r + 1) & 7]), "+v"(a[(r + 2) & 7])); }
                    }
                    if (KIND == K_MIX) {
                        // SHA-256's proportions per 8 instructions: 3 rotates, 2 three-input logic ops, 1 add3, 1 add, 1 shift
                        if (i < 3) a[i] = __builtin_amdgcn_alignbit(a[i], a[i], 7 + i);
                        else if (i < 5) a[i] = __builtin_amdgcn_bitop3_b32(a[i], a[(i + 1) & 7], a[(i + 2) & 7], 0x96);
                        else if (i == 5) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                        else if (i == 6) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                        else asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
                    }
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63u) == 0u) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        o[0] = t0; o[1] = r0; o[2] = t1; o[3] = r1;
    }
}

// ---- sysfs hwmon sampling ----------------------------------------------------------
static std::string first_glob(const char* pat)
{
    glob_t g;
    std::string r;
    if (glob(pat, 0, nullptr, &g) == 0 && g.gl_pathc > 0) r = g.gl_pathv[0];
    globfree(&g);
    return r;
}
static double read_num(const std::string& p)
{
    if (p.empty()) return -1;
    FILE* f = fopen(p.c_str(), "r");
    if (!f) return -1;
    double v = -1;
    if (fscanf(f, "%lf", &v) != 1) v = -1;
    fclose(f);
    return v;
}
struct Hwmon {
    std::string power, cap, sclk;
    Hwmon()
    {
        // the host has several GPUs: take the hwmon node of the PCI function HIP device 0 sits on
        char id[64] = "";
        std::string base = "/sys/class/drm/card*/device/hwmon/hwmon*/";
        if (hipDeviceGetPCIBusId(id, sizeof id, 0) == hipSuccess && id[0]) {
            for (char* c = id; *c; ++c) *c = (char)tolower(*c);
            base = std::string("/sys/bus/pci/devices/") + id + "/hwmon/hwmon*/";
        }
        power = first_glob((base + "power1_average").c_str());
        if (power.empty()) power = first_glob((base + "power1_input").c_str());
        cap = first_glob((base + "power1_cap").c_str());
        sclk = first_glob((base + "freq1_input").c_str());
    }
};
static double median(std::vector<double> v)
{
    if (v.empty()) return -1;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

template <int KIND>
void run(const char* name, int waves_per_simd, int iters, uint32_t data_mask, double ops_per_lane_iter, double slots_per_lane_iter,
         uint32_t* d_out, unsigned long long* d_st, const Hwmon& hw, double warm_s)
{
    const int blocks = 256 * waves_per_simd;   // 256-lane workgroups = one wavefront per SIMD each
    std::atomic<bool> stop{false};
    std::vector<double> pw, sk;
    std::thread sampler([&] {
        while (!stop.load()) {
            const double p = read_num(hw.power), s = read_num(hw.sclk);
            if (p > 0) pw.push_back(p / 1e6);
            if (s > 0) sk.push_back(s / 1e6);
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    });
    // >= warm_s of back-to-back launches, then the measured one
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    do {
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters, data_mask);
        (void)hipDeviceSynchronize();
        launches += 8;
    } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < warm_s);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters, data_mask);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    stop.store(true);
    sampler.join();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st((size_t)blocks * 4 * 4);
    (void)hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (size_t w = 0; w < (size_t)blocks * 4; ++w) {
        const double dt = (double)(st[4 * w + 2] - st[4 * w]), dr = (double)(st[4 * w + 3] - st[4 * w + 1]);
        if (dr > 0) { clk.push_back(dt / dr * 0.1); cyc.push_back(dt); }   // GHz
    }
    const double ghz = median(clk);
    std::sort(clk.begin(), clk.end());
    const double lanes = (double)blocks * 256;
    const double tops = lanes * iters * ops_per_lane_iter / (ms * 1e-3) / 1e12;
    const double tslots = lanes * iters * slots_per_lane_iter / (ms * 1e-3) / 1e12;
    const double peak_at_clock = 256.0 * 4 * 32 * ghz * 1e9 / 1e12;   // full-rate lane-ops/s at the measured clock
    // per-SIMD issue: each SIMD hosts waves_per_simd wavefronts for the whole launch; a wavefront's loop takes
    // median(cyc) cycles and issues iters x slots_per_lane_iter full-rate-equivalent slots of 2 cycles each
    const double busy = waves_per_simd * iters * slots_per_lane_iter * 2.0 / median(cyc);
    printf("%-22s %d w/SIMD data=%s  %8.3f ms  %6.2f T ops/s  %6.2f T slots/s | in-kernel clock %.3f GHz (p5 %.3f, p95 %.3f)"
           " | slots/s = %.3f of the issue peak at that clock (%.1f T), in-loop issue occupancy %.3f | sclk(sysfs) %.0f MHz, power %.0f W (cap %.0f W), %d launches warm\n",
           name, waves_per_simd, data_mask ? "random" : "zeros ", ms, tops, tslots, ghz, clk[clk.size() / 20], clk[clk.size() - 1 - clk.size() / 20],
           tslots / peak_at_clock, peak_at_clock, busy, median(sk), median(pw), read_num(hw.cap) / 1e6, launches);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const double warm_s = argc > 1 ? atof(argv[1]) : 2.0;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("device: %s arch=%s CUs=%d clockRate=%d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    Hwmon hw;
    printf("hwmon: power=%s cap=%s sclk=%s\n", hw.power.c_str(), hw.cap.c_str(), hw.sclk.c_str());
    uint32_t* d_out;
    unsigned long long* d_st;
    (void)hipMalloc(&d_out, (size_t)256 * 8 * 256 * 4);
    (void)hipMalloc(&d_st, (size_t)256 * 8 * 4 * 4 * 8);
    // slots: full-rate instructions 1, half-rate (alignbit, add3) 2 -- profiles/r01_valu_issue_rates.txt
    const double node_slots = 5708.0;   // tools/isa_count.py on reduce_level_kernel: one hash_pair
    const bool full = argc > 2 && atoi(argv[2]) == 1;
    for (int w : {2, 4, 8}) {
        const int it = 16000 / w;
        // (single-opcode streams of asm statements get an s_nop from hipcc after most of them: those live in
        //  tools/issue_patterns.hip, one asm block per loop body; here only what the compiler emits by itself)
        run<K_ALIGNBIT>("v_alignbit_b32 d8", w, it / 2, 0xffffffffu, 128, 256, d_out, d_st, hw, warm_s);
        run<K_BITOP3>("v_bitop3_b32 d8", w, it, 0xffffffffu, 128, 128, d_out, d_st, hw, warm_s);
        run<K_MIX>("sha-like mix", w, it / 2, 0xffffffffu, 128, 16 * (3 * 2 + 2 + 2 + 1 + 1), d_out, d_st, hw, warm_s);
        run<K_NODE>("node hash (hash_pair)", w, 512 / w, 0xffffffffu, 1, node_slots, d_out, d_st, hw, warm_s);
        run<K_NODE2>("2 node hashes per lane", w, 256 / w, 0xffffffffu, 2, 2 * node_slots, d_out, d_st, hw, warm_s);
        if (full) run<K_NODE>("node hash (hash_pair)", w, 512 / w, 0u, 1, node_slots, d_out, d_st, hw, warm_s);
    }
    return 0;
}
