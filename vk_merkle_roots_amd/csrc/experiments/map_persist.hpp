// map_persist.hpp -- the LDS-staged MAP kernel as PERSISTENT workgroups (gfx950): tile k+1 is on its way from HBM while
// tile k is hashed.  Same tiles, same sort, same staging and the same hashing as map_kernel's MODE 0 (map_kernel.hpp; reference
// shader src/shaders/SHA-256.comp:177-304) -- what changes is when the bytes travel:
//   * a workgroup takes tiles b, b + G, b + 2G ... (G = gridDim.x = two workgroups per CU) instead of one -- at any moment the
//     chip works on G neighbouring tiles, so metadata, bytes and digests spread over the HBM channels as they do with one tile
//     per workgroup (runs of consecutive tiles put every workgroup's 1 MiB-strided metadata and 4 MiB-strided digests on the
//     same channels: 7.6 ms against 5.4, profiles/r04_map_persist.txt);
//   * at the start of tile k's hashing every lane requests its share of tile k+1: its metadata entries (PER x 8 bytes) and
//     its words of the packed bytes (NW = 35 single words: nine 16-byte tuples held across the hashing find no aligned homes
//     and are spilled) into registers -- 43 VGPRs the hashing does not need at four wavefronts per SIMD.  Both HBM round trips of map_kernel's prologue (metadata, then the bytes whose extent the metadata
//     gives) run under ~35 us of hashing;
//   * the extent of the next tile's bytes is taken from its first and last metadata entries alone (two loads, requested one
//     tile earlier still), which is exact for a packed batch (every string starts on the word after the one before,
//     Batches.cpp:64-121).  Every lane then checks that ITS entries lie inside that extent (one ballot per wavefront); when any
//     does not (permuted or corrupt metadata) the registers are dropped, the true extent is computed from all entries as
//     map_kernel does and the tile is staged the old way.  Nothing is trusted that is not checked.
// The loads are issued and consumed inside ONE iteration of the tile loop (issue, hash the tile before, consume): carried
// around the loop's back edge they would be copied register to register, and a copy waits for the load.
// gfx950 retires vector memory loads in order under ONE counter (vmcnt), and the compiler's wait insertion merges "may be
// pending" over every path of a function: with other loads it can see anywhere in the tile loop it puts `s_waitcnt vmcnt(3)`
// in front of innocent register writes in the hashing loop (registers one of those loads might have been written to), and
// such a wait also waits for every OLDER load -- the whole next tile.  So the tile loop holds no other load the compiler can
// see: the per-lane loads of a tile that could not be staged are spelled as assembly with their own waits (each wait names
// the registers it makes valid, so nothing that uses them can move above it), and the hashing loop of a staged tile issues
// no vector memory load at all.  The prefetch itself is ordinary C++: the compiler knows those registers are in flight, never
// reads them early, and waits exactly where install() first uses them.  (Tried and dropped: the prefetch as assembly into
// accumulation registers -- with AGPRs in play the allocator halves the architectural budget, spills into AGPRs itself, runs
// out of them and stores not-yet-arrived tuples to scratch.)
// Between tiles the workgroup passes three barriers (tile done -> entries in LDS and counted -> order and bytes in LDS); no HBM
// latency sits between them.  The counting sort ranks by ballot -- one LDS atomic per wavefront and distinct block count
// instead of one per string on two or three hot addresses -- and every wavefront turns the histogram into bin starts itself.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../map_kernel.hpp"

typedef uint32_t vkmr_u32x4 __attribute__((ext_vector_type(4)));

// Loads the compiler does not see (see the header).  The value is valid only behind one of the waits below.
__device__ __forceinline__ void unseen_load_b128(vkmr_u32x4& v, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p)); }
__device__ __forceinline__ void unseen_load_b32(uint32_t& v, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p)); }
__device__ __forceinline__ void unseen_wait(vkmr_u32x4& a, vkmr_u32x4& b, vkmr_u32x4& c, vkmr_u32x4& d)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// One group of 64 sorted strings of the tile in LDS (map_kernel's MODE 0 body): words from the staging area, or -- a tile that
// could not be staged -- per lane from HBM.
template <bool STAGED>
__device__ __forceinline__ void hash_group(const uint32_t* __restrict__ data, uint64_t data_words, Node* __restrict__ out_tile, const uint2* s_meta,
                                           const uint16_t* s_order, const uint32_t* s_stage, uint32_t g, uint32_t n_tile, uint32_t lane,
                                           unsigned long long a0, uint32_t span)
{
    const uint32_t pos = g * 64u + lane;
    const bool has = pos < n_tile;
    const uint32_t id = has ? s_order[pos] : 0u;
    const uint2 md = s_meta[id];
    const uint32_t start = md.x, size = has ? md.y : 0u;
    const uint32_t nb = has ? block_count(size) : 0u;
    uint32_t H[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) H[i] = vkmr_dev::IV256[i];
    for (uint32_t b = 0; __any(b < nb); ++b) {
        uint32_t w[16];
        if (STAGED) {
            uint32_t base = (uint32_t)(start - a0) + (b << 4);
            base = base < span ? base : span;
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = s_stage[base + i];
        } else {
            const uint64_t gbase = (uint64_t)start + ((uint64_t)b << 4);
            if (gbase + 16u <= data_words) {   // strings are 4-byte aligned; gfx950 takes dword-aligned dwordx4
                vkmr_u32x4 x0, x1, x2, x3;
                const uint32_t* src = data + gbase;
                unseen_load_b128(x0, src); unseen_load_b128(x1, src + 4); unseen_load_b128(x2, src + 8); unseen_load_b128(x3, src + 12);
                unseen_wait(x0, x1, x2, x3);
                w[0] = x0.x; w[1] = x0.y; w[2] = x0.z; w[3] = x0.w; w[4] = x1.x; w[5] = x1.y; w[6] = x1.z; w[7] = x1.w;
                w[8] = x2.x; w[9] = x2.y; w[10] = x2.z; w[11] = x2.w; w[12] = x3.x; w[13] = x3.y; w[14] = x3.z; w[15] = x3.w;
            } else if (data_words != 0) {   // the buffer's end: word by word
                // Every load is issued by every lane, from an index clamped into the buffer: words beyond the buffer lie beyond the
                // string (sizes are cut at the buffer's end) and are masked below like any byte beyond `size`.  No load sits under a
                // condition of its own -- a value merged from "loaded" and "not loaded" lanes would be copied before the wait.
                const uint64_t last = data_words - 1u;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint64_t idx = gbase + i;
                    unseen_load_b32(w[i], data + (idx < last ? idx : last));
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]),
                                                    "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]));
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) w[i] = 0u;
            }
        }
        const uint64_t boff = (uint64_t)b << 6;
        const uint32_t r = (boff >= size) ? 0u : ((size - boff >= 64u) ? 64u : (uint32_t)(size - boff));
        uint32_t term = ((boff <= size) && (size - boff < 64u)) ? 0xFFFFFFFFu : 0u;   // the 0x80 byte falls in this block
        asm("" : "+v"(term));
        const uint32_t kb = (r & 3u) << 3;
        const uint32_t keep = ~(0xFFFFFFFFu >> kb);
        const uint32_t padbit = 0x80000000u >> kb;
        const uint32_t full = r >> 2;
        uint32_t M[16];
        whole_word_masks(full, M);
        uint32_t prev = term;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t x = __builtin_bswap32(w[i]);
            const uint32_t bnd = __builtin_amdgcn_bitop3_b32(x, keep, padbit, 0xEA);
            const uint32_t u = bnd & prev;
            w[i] = __builtin_amdgcn_bitop3_b32(x, M[i], u, 0xE2);
            prev = M[i] & term;
        }
        if (b + 1u == nb) {   // last block carries the 64-bit bit length (CPU path, SHA-256plus.cpp:100-117)
            w[14] = size >> 29;
            w[15] = size << 3;
        }
        if (b < nb) vkmr_dev::compress(H, w);
    }
    if (has) {
        uint32_t o[8];
        vkmr_dev::hash_digest(H, o);
        vkmr_dev::store_node(out_tile + id, o);
    }
}

// Cross-lane reads with the source lane's byte address computed HERE from the caller's (per-tile, opaque) lane index: the
// library's __shfl forms derive it from the hardware lane id, which the compiler hoists out of the tile loop and then keeps --
// a dozen registers of shuffle addresses held across the hashing.
__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }
__device__ __forceinline__ unsigned long long lane_read64(unsigned long long v, uint32_t src_lane)
{
    return ((unsigned long long)lane_read((uint32_t)(v >> 32), src_lane) << 32) | lane_read((uint32_t)v, src_lane);
}

template <int THREADS, int MAX_TILE, int STAGE_WORDS>
__global__ __launch_bounds__(THREADS, 4) void map_persist_kernel(const uint32_t* __restrict__ data, uint64_t data_words,
                                                                 const vkmr_metadata* __restrict__ meta, uint32_t count,
                                                                 Node* __restrict__ out, uint32_t tile, uint32_t ntiles, uint32_t stagger)
{
    __shared__ uint4 s_stage4[(STAGE_WORDS + VKMR_MAP_STAGE_PAD) / 4];
    __shared__ uint2 s_meta[MAX_TILE];
    __shared__ uint16_t s_order[MAX_TILE];
    __shared__ uint32_t s_hist[VKMR_MAP_BINS];
    __shared__ unsigned long long s_lo, s_hi;
    __shared__ uint32_t s_next, s_outside;
    uint32_t* s_stage = reinterpret_cast<uint32_t*>(s_stage4);

    constexpr int PER = MAX_TILE / THREADS;
    constexpr int NV = (STAGE_WORDS / 4 + THREADS - 1) / THREADS;   // 16-byte pieces per lane (the staging of a tile that was not sent ahead)
    constexpr int NW = (STAGE_WORDS + THREADS - 1) / THREADS;        // words per lane sent ahead: single registers -- nine 4-register tuples held
                                                                     // across the hashing find no aligned homes and are spilled to scratch
    const bool aligned = (reinterpret_cast<uintptr_t>(data) & 15u) == 0u;
    const uint2* meta2 = reinterpret_cast<const uint2*>(meta);

    uint32_t t = blockIdx.x;
    if (t >= ntiles) return;
    // The two workgroups of a CU start together and their tiles cost the same: left alone they install together and hash
    // together, tile after tile -- the CU idles through every install twice over (5.85 ms against 5.3, profiles/r04_map_persist.txt).
    // The second half of the grid (dispatched onto CUs that already hold a workgroup) starts half a tile late.
    if (stagger && blockIdx.x >= gridDim.x / 2u)
        for (uint32_t k = 0; k < stagger; ++k) __builtin_amdgcn_s_sleep(127);   // 127 x 64 cycles each
    VKMR_STAMP(t_begin);
    VKMR_STAMP_RT(rt_begin);
#ifdef VKMR_STAMPS
    unsigned long long acc_wait = 0, acc_sort = 0, acc_stage = 0;
#endif

    auto tile_count = [&](uint32_t tt) -> uint32_t {
        const uint64_t base = (uint64_t)tt * tile;
        return (uint32_t)((count - base < tile) ? count - base : tile);
    };

    // what the hashing of the current tile needs to know about its staging (uniform)
    unsigned long long a0 = 0;
    uint32_t span = 0;
    bool staged = false;

    // Tile tt from registers (or, when `have` is false or an entry lies outside the requested extent, from HBM) into LDS:
    // entries, order, bytes.  Entered by every wavefront once it has finished the tile before.  [spec_a0, spec_hi): the
    // words the registers were requested for (spec_a0 a multiple of 4, the whole span inside the buffer and the staging area).
    auto install = [&](uint32_t tt, const uint2 (&mdv)[PER], const uint32_t (&pre)[NW], bool have, unsigned long long spec_a0, unsigned long long spec_hi) {
        const uint32_t n_tile = tile_count(tt);
        // The lane's index, opaque per tile: everything derived from it (LDS and HBM offsets, bounds tests) is recomputed here
        // instead of being hoisted out of the tile loop and held -- spilled -- across the hashing.
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const uint32_t lane = tid & 63u;
        VKMR_STAMP(t0);
        __builtin_amdgcn_s_setprio(3);
        __syncthreads();             // every wavefront has finished the tile before: LDS is free (also publishes the counters' reset)
        VKMR_STAMP(t1);
        if (tid == 0) s_next = 0u;
        uint32_t key[PER], rank[PER];
        unsigned long long lo = ~0ull, hi = 0ull;
        bool inside = true;
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t i = tid + k * THREADS;
            key[k] = 0u; rank[k] = 0u;
            const bool valid = i < n_tile;
            if (valid) {
                uint2 md = mdv[k];
                const unsigned long long avail = (md.x < data_words) ? (data_words - md.x) * 4ull : 0ull;   // cut at the buffer's end (map_kernel)
                md.y = (md.y > avail) ? (uint32_t)avail : md.y;
                s_meta[i] = md;
                const uint32_t nb = block_count(md.y);
                key[k] = nb < VKMR_MAP_BINS ? nb : (VKMR_MAP_BINS - 1u);
                const unsigned long long b = md.x, e = b + (((unsigned long long)md.y + 3ull) >> 2);
                lo = b < lo ? b : lo;
                hi = e > hi ? e : hi;
                inside = inside && b >= spec_a0 && e <= spec_hi;
            }
            // rank among the tile's strings of the same block count: by ballot inside the wavefront, one atomic per wavefront
            // and distinct count (two or three for strings like rndm's; 63 at most)
            unsigned long long todo = __ballot(valid);
            while (todo) {
                const uint32_t kk = __builtin_amdgcn_readlane(key[k], (int)__builtin_ctzll(todo));
                const unsigned long long same = __ballot(valid && key[k] == kk);
                uint32_t base = 0u;
                if (lane == (uint32_t)__builtin_ctzll(same)) base = atomicAdd(&s_hist[kk], (uint32_t)__builtin_popcountll(same));
                base = __builtin_amdgcn_readlane(base, (int)__builtin_ctzll(same));
                if (valid && key[k] == kk) rank[k] = base + (uint32_t)__builtin_popcountll(same & lt_mask);
                todo &= ~same;
            }
        }
        // The extent: with registers in hand, "every entry lies inside what was requested" is all that is needed (a ballot);
        // without, or when an entry does not, the minimum and maximum over the tile as in map_kernel.
        const bool wave_inside = have && __all(inside);
        if (!wave_inside) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const unsigned long long ol = lane_read64(lo, lane ^ (uint32_t)d), oh = lane_read64(hi, lane ^ (uint32_t)d);
                lo = ol < lo ? ol : lo;
                hi = oh > hi ? oh : hi;
            }
            if (lane == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); if (have) s_outside = 1u; }
        }
        __syncthreads();
        bool from_regs = have && s_outside == 0u;
        if (have && !from_regs) {
            // some wavefront found an entry outside: the others have not contributed their extremes yet
            if (wave_inside) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const unsigned long long ol = lane_read64(lo, lane ^ (uint32_t)d), oh = lane_read64(hi, lane ^ (uint32_t)d);
                    lo = ol < lo ? ol : lo;
                    hi = oh > hi ? oh : hi;
                }
                if (lane == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
            }
            __syncthreads();
        }
        // bin starts (longest first): every wavefront scans the histogram itself -- lane j takes bin j
        uint32_t excl;
        {
            const uint32_t mine = s_hist[lane];
            uint32_t x = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = lane_read(x, (lane + (uint32_t)d) & 63u);
                x += (lane + d < 64u) ? y : 0u;
            }
            excl = x - mine;   // strings in bins above this one
        }
        VKMR_STAMP(t2);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t i = tid + k * THREADS;
            const uint32_t start_of_bin = lane_read(excl, key[k]);
            if (i < n_tile) s_order[start_of_bin + rank[k]] = (uint16_t)i;
        }
        if (from_regs) {
            a0 = spec_a0;
            span = (uint32_t)(spec_hi - spec_a0);
            staged = true;
            // the registers hold exactly these words
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const uint32_t w = tid + q * THREADS;
                if (w < span) s_stage[w] = pre[q];
            }
        } else {
            const unsigned long long t_lo = s_lo, t_hi = s_hi;
            a0 = t_lo & ~3ull;
            staged = (t_hi >= t_lo) && (t_hi - a0 <= (unsigned long long)STAGE_WORDS) && (t_hi <= data_words) && aligned;
            span = staged ? (uint32_t)(t_hi - a0) : 0u;
            if (staged) {
                // not in the registers (first tile, metadata not ascending): fetch it now, as map_kernel does
                const uint4* src4 = reinterpret_cast<const uint4*>(data + a0);
                uint4 v[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const uint32_t w = (tid + q * THREADS) * 4u;
                    v[q] = make_uint4(0u, 0u, 0u, 0u);
                    if (w < span) {
                        if (a0 + w + 4u <= data_words) {
                            v[q] = src4[w >> 2];
                        } else {   // last, partial vector of the buffer
                            v[q].x = data[a0 + w];
                            v[q].y = (a0 + w + 1u < data_words) ? data[a0 + w + 1u] : 0u;
                            v[q].z = (a0 + w + 2u < data_words) ? data[a0 + w + 2u] : 0u;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const uint32_t w = (tid + q * THREADS) * 4u;
                    if (w < span) s_stage4[w >> 2] = v[q];
                }
            }
        }
        __syncthreads();
        // the counters' next use is behind the next tile's first barrier; every wavefront has read them
        if (tid < VKMR_MAP_BINS) s_hist[tid] = 0u;
        if (tid == 0) { s_lo = ~0ull; s_hi = 0ull; s_outside = 0u; }
        __builtin_amdgcn_s_setprio(0);
        VKMR_STAMP(t3);
#ifdef VKMR_STAMPS
        acc_wait += t1 - t0; acc_sort += t2 - t1; acc_stage += t3 - t2;
#endif
    };

    // The words tile tt will occupy if its first and last entries tell the truth (uniform): [lo & ~3, hi); valid when that
    // is a span the staging area and the buffer hold in whole 16-byte pieces.
    auto extent_of = [&](uint2 first, uint2 last, unsigned long long& e_a0, unsigned long long& e_hi) -> bool {
        const unsigned long long lo = first.x, hi = (unsigned long long)last.x + (((unsigned long long)last.y + 3ull) >> 2);
        e_a0 = lo & ~3ull;
        e_hi = hi;
        return aligned && hi >= lo && hi - e_a0 <= (unsigned long long)STAGE_WORDS && e_a0 + ((hi - e_a0 + 3ull) & ~3ull) <= data_words;
    };

    if (threadIdx.x < VKMR_MAP_BINS) s_hist[threadIdx.x] = 0u;
    if (threadIdx.x == 0) { s_lo = ~0ull; s_hi = 0ull; s_outside = 0u; }
    // Which tiles: b, b + G, b + 2G ...  (Tried: every tile after the first three taken by ticket, one atomic per tile -- the launch
    // then ends when the work does instead of waiting for its slowest workgroup, 5.28 ms instead of 5.85, which is map_kernel's
    // 5.2 ms and no better: profiles/r04_map_persist.txt.)
    // the extent the NEXT tile's end entries promise, known one tile ahead (uniform; carried around the loop as plain numbers)
    unsigned long long next_a0 = 0, next_hi = 0;
    bool next_ok = false;
    {   // the first tile pays both round trips, as every tile of map_kernel does
        uint2 mdv[PER];
        uint32_t none[NW];
        const uint64_t base = (uint64_t)t * tile;
        const uint32_t n = tile_count(t);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t i = threadIdx.x + k * THREADS;
            mdv[k] = make_uint2(0u, 0u);
            if (i < n) mdv[k] = meta2[base + i];
        }
        const uint32_t tn = t + gridDim.x;
        uint2 first = make_uint2(0u, 0u), last = make_uint2(0u, 0u);
        if (tn < ntiles) {
            first = meta2[(uint64_t)tn * tile];
            last = meta2[(uint64_t)tn * tile + tile_count(tn) - 1u];
        }
        install(t, mdv, none, false, 0ull, 0ull);
        if (tn < ntiles) next_ok = extent_of(first, last, next_a0, next_hi);
    }

    for (;;) {
        const uint64_t tile_base = (uint64_t)t * tile;
        const uint32_t n_tile = tile_count(t);
        const uint32_t lane = threadIdx.x & 63u;
        // ---- the next tile sets out ----------------------------------------------------------------------------------------
        const uint32_t tn = t + gridDim.x, tnn = tn + gridDim.x;
        const bool more = tn < ntiles;
        // the next tile's entries and bytes, and the end entries of the tile after it.  Not initialised: a write would have to wait
        // for the load the register received a tile ago; lanes that load nothing never look at theirs.
        uint2 mdv[PER], e_first, e_last;
        uint32_t pre[NW];
        const unsigned long long spec_a0 = next_a0, spec_hi = next_hi;
        // (a tile that could not be staged hashes with per-lane loads of its own: nothing is sent ahead under those -- their
        // registers on top of the bytes in flight are more than a wavefront has)
        const bool have = more && next_ok && staged;
        auto request_entries = [&] {
            uint32_t tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const uint64_t base = (uint64_t)tn * tile;
            const uint32_t n = tile_count(tn);
            if (tnn < ntiles) {
                // every lane the same two addresses -- through a vector register on purpose: a scalar load shares its counter with
                // the LDS reads of the hashing loop, whose first wait would then sit out this load's trip to HBM
                uint32_t zero = 0u;
                asm volatile("" : "+v"(zero));
                e_first = meta2[(uint64_t)tnn * tile + zero];
                e_last = meta2[(uint64_t)tnn * tile + tile_count(tnn) - 1u + zero];
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const uint32_t i = tid + k * THREADS;
                if (i < n) mdv[k] = meta2[base + i];
            }
        };

        // ---- hash groups of 64 sorted strings (map_kernel, MODE 0) -----------------------------------------------------------
        // Two loops, one per source of the words: the staged one issues no vector memory load at all, so nothing in it can make
        // the in-order load counter wait for the next tile's bytes.
        const uint32_t ngroups = (n_tile + 63u) >> 6;
        if (more) request_entries();
        if (staged) {
            if (have) {
                uint32_t tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                const uint32_t* src = data + spec_a0;
                const uint32_t sp = (uint32_t)(spec_hi - spec_a0);
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    const uint32_t w = tid + q * THREADS;
                    if (w < sp) pre[q] = src[w];
                }
            }
            for (;;) {
                uint32_t g = 0u;
                if (lane == 0) g = atomicAdd(&s_next, 1u);
                g = __builtin_amdgcn_readfirstlane(g);
                if (g >= ngroups) break;
                hash_group<true>(data, data_words, out + tile_base, s_meta, s_order, s_stage, g, n_tile, lane, a0, span);
            }
        } else {   // a tile that could not be staged: per-lane loads
            for (;;) {
                uint32_t g = 0u;
                if (lane == 0) g = atomicAdd(&s_next, 1u);
                g = __builtin_amdgcn_readfirstlane(g);
                if (g >= ngroups) break;
                hash_group<false>(data, data_words, out + tile_base, s_meta, s_order, s_stage, g, n_tile, lane, a0, span);
            }
        }
        if (!more) break;
        // ---- the tile in the registers becomes the tile in LDS -------------------------------------------------------------
        next_ok = false;
        if (tnn < ntiles) {
            const uint2 first = make_uint2(__builtin_amdgcn_readfirstlane(e_first.x), __builtin_amdgcn_readfirstlane(e_first.y));
            const uint2 last = make_uint2(__builtin_amdgcn_readfirstlane(e_last.x), __builtin_amdgcn_readfirstlane(e_last.y));
            next_ok = extent_of(first, last, next_a0, next_hi);
        }
        install(tn, mdv, pre, have, spec_a0, spec_hi);
        t = tn;
    }
#ifdef VKMR_STAMPS
    {
        unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0 && blockIdx.x < VKMR_STAMP_SLOTS) {   // wavefront 0 of each workgroup: time by phase, summed over its tiles
            unsigned long long* o = g_stamps + (size_t)blockIdx.x * 8;
            o[0] = t_begin; o[1] = rt_begin; o[2] = t_begin + acc_sort; o[3] = t_begin + acc_sort + acc_stage; o[4] = t_end; o[5] = rt_end;
            o[6] = 0x4d4150ull /* "MAP" */ | (acc_wait << 24);   // bits 24..63: cycles spent waiting at the tile-done barrier (part of "hash" above)
            o[7] = gridDim.x;
        }
    }
#endif
}
