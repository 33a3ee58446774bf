/*
 * sha256d_oracle.c -- CPU restatement of the reference's serial Merkle-root path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker.  The product path
 * (include/vkmr_hip.h, vk_merkle_roots_amd/) never links or calls it.
 *
 * Parity status: PINNED.  The reference repository holds no tests or golden
 * vectors of its own, so this restatement is pinned
 * against (1) the reference's own CPU-serial sources compiled unmodified into
 * oracle/_ref/ by oracle/Makefile and run on the same streams, and (2) the
 * fixtures under tests/golden/ that were produced by that build
 * (tests/golden/make_golden.py).  tests/test_oracle.py checks both.
 *
 * Each function cites the reference lines (relative to /root/reference) whose
 * behaviour it restates.  The code is written from the algorithm, not from
 * the reference text: one generic streaming SHA-256 core replaces the
 * reference's three hand-specialised block builders.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define ORACLE_API __attribute__((visibility("default")))

/* FIPS 180-4 round constants; same table as src/vkmr/SHA-256plus.cpp:76-97. */
static const uint32_t K256[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u
};

/* Initial hash value, src/vkmr/SHA-256plus.cpp:122-131. */
static const uint32_t IV256[8] = {
    0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u
};

static inline uint32_t rotr32(uint32_t x, unsigned n) { return (x >> n) | (x << (32u - n)); }

/*
 * One compression of a 16-word big-endian-valued block into state H.
 * Restates the round loop that appears three times in the reference:
 * src/vkmr/SHA-256plus.cpp:229-266 (leaf), :322-344 (digest), :404-433 (pair),
 * with the macros of src/common/SHA-256defs.h:16-26.
 */
static void compress(uint32_t H[8], const uint32_t M[16])
{
    uint32_t W[64];
    for (int t = 0; t < 16; ++t) W[t] = M[t];
    for (int t = 16; t < 64; ++t) {
        uint32_t s0 = rotr32(W[t - 15], 7) ^ rotr32(W[t - 15], 18) ^ (W[t - 15] >> 3);
        uint32_t s1 = rotr32(W[t - 2], 17) ^ rotr32(W[t - 2], 19) ^ (W[t - 2] >> 10);
        W[t] = s1 + W[t - 7] + s0 + W[t - 16];
    }
    uint32_t a = H[0], b = H[1], c = H[2], d = H[3], e = H[4], f = H[5], g = H[6], h = H[7];
    for (int t = 0; t < 64; ++t) {
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t T1 = h + S1 + ch + K256[t] + W[t];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t T2 = S0 + mj;
        h = g; g = f; f = e; e = d + T1; d = c; c = b; b = a; a = T1 + T2;
    }
    H[0] += a; H[1] += b; H[2] += c; H[3] += d; H[4] += e; H[5] += f; H[6] += g; H[7] += h;
}

/*
 * SHA-256 of an arbitrary byte string; result as eight word VALUES H[0..7]
 * (the representation of VkSha256Result, src/common/SHA-256defs.h:47-49, and of
 * the reference's vector<uint32_t> nodes).  Restates cpu_sha256_n,
 * src/vkmr/SHA-256plus.cpp:119-276: block count incl. the extra block when fewer
 * than 9 bytes remain (:138-149), 0x80 terminator (:160-166, :203-216), 64-bit
 * big-endian bit length in the last two words (:100-117).
 */
static void sha256_words(const uint8_t *msg, size_t len, uint32_t out[8])
{
    uint32_t H[8];
    memcpy(H, IV256, sizeof H);
    size_t full = len / 64;
    uint32_t M[16];
    for (size_t i = 0; i < full; ++i) {
        const uint8_t *p = msg + 64 * i;
        for (int w = 0; w < 16; ++w)
            M[w] = ((uint32_t)p[4 * w] << 24) | ((uint32_t)p[4 * w + 1] << 16) | ((uint32_t)p[4 * w + 2] << 8) | p[4 * w + 3];
        compress(H, M);
    }
    uint8_t tail[128];
    size_t rem = len - 64 * full;
    memset(tail, 0, sizeof tail);
    if (rem) memcpy(tail, msg + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = (rem + 1 + 8 <= 64) ? 64 : 128;
    uint64_t bits = (uint64_t)len << 3;
    for (int i = 0; i < 8; ++i) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    for (size_t off = 0; off < tl; off += 64) {
        const uint8_t *p = tail + off;
        for (int w = 0; w < 16; ++w)
            M[w] = ((uint32_t)p[4 * w] << 24) | ((uint32_t)p[4 * w + 1] << 16) | ((uint32_t)p[4 * w + 2] << 8) | p[4 * w + 3];
        compress(H, M);
    }
    memcpy(out, H, sizeof H);
}

/*
 * SHA-256 of a 32-byte digest given as word values.  Restates cpu_sha256_1,
 * src/vkmr/SHA-256plus.cpp:278-358: M[0..7]=digest, M[8]=0x80000000, M[15]=256.
 */
static void sha256_of_digest(const uint32_t in[8], uint32_t out[8])
{
    uint32_t M[16] = {0};
    memcpy(M, in, 32);
    M[8] = 0x80000000u;
    M[15] = 256u;
    uint32_t H[8];
    memcpy(H, IV256, sizeof H);
    compress(H, M);
    memcpy(out, H, sizeof H);
}

/*
 * SHA-256 of left||right (64 bytes) given as word values.  Restates cpu_sha256_2,
 * src/vkmr/SHA-256plus.cpp:360-451: first block = the two digests, second block
 * = 0x80000000, zeros, 512 (:441-447).
 */
static void sha256_of_pair(const uint32_t l[8], const uint32_t r[8], uint32_t out[8])
{
    uint32_t M[16];
    memcpy(M, l, 32);
    memcpy(M + 8, r, 32);
    uint32_t H[8];
    memcpy(H, IV256, sizeof H);
    compress(H, M);
    memset(M, 0, sizeof M);
    M[0] = 0x80000000u;
    M[15] = 512u;
    compress(H, M);
    memcpy(out, H, sizeof H);
}

/* Plain SHA-256 to canonical bytes; cpu_sha256, src/vkmr/SHA-256plus.cpp:473-477,
 * with hash_to_string's byte order (:453-469). */
ORACLE_API void oracle_sha256(const uint8_t *msg, size_t len, uint8_t out[32])
{
    uint32_t h[8];
    sha256_words(msg, len, h);
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
    }
}

/* Leaf digest = SHA-256(SHA-256(bytes)) as word values; cpu_sha256d_int,
 * src/vkmr/SHA-256plus.cpp:479 and CpuSha256D::Add (:558-561). */
ORACLE_API void oracle_leaf(const uint8_t *msg, size_t len, uint32_t out[8])
{
    uint32_t h[8];
    sha256_words(msg, len, h);
    sha256_of_digest(h, out);
}

/* Tree node = SHA-256d(l||r); the body of CpuSha256D::Root's pair loop,
 * src/vkmr/SHA-256plus.cpp:528-530. */
ORACLE_API void oracle_node(const uint32_t l[8], const uint32_t r[8], uint32_t out[8])
{
    uint32_t h[8];
    sha256_of_pair(l, r, h);
    sha256_of_digest(h, out);
}

/*
 * Duplicate-last Merkle root over n >= 1 word-valued leaves, IN PLACE (the
 * reference also destroys its leaf vector).  Restates CpuSha256D::Root,
 * src/vkmr/SHA-256plus.cpp:491-556: do { pairs = ceil(count/2); r = (no right
 * sibling ? l : right) } while (size > 1) -- so a single leaf is hashed with
 * itself once.  Returns 0, or -1 when n == 0 (the reference
 * returns "" there, :494-496).
 */
ORACLE_API int oracle_root_inplace(uint32_t *nodes, size_t n, uint32_t out[8])
{
    if (n == 0) return -1;
    do {
        size_t pairs = (n + 1) / 2;
        for (size_t p = 0; p < pairs; ++p) {
            const uint32_t *l = nodes + 16 * p;
            const uint32_t *r = (2 * p + 1 < n) ? l + 8 : l;
            uint32_t h[8];
            oracle_node(l, r, h);
            memcpy(nodes + 8 * p, h, 32);
        }
        n = pairs;
    } while (n > 1);
    memcpy(out, nodes, 32);
    return 0;
}

/*
 * Sub-tree root of one slice reduced through exactly `height` levels with the
 * duplicate-last rule at every level, continuing with self-pairing after the
 * count has collapsed to one.  This is the per-slice contract of the device
 * reduction: applicable = (slice# > 1 ? Capacity : Count) levels,
 * src/vkmr/Reductions.cpp:471-472 and README.md:94; pairing rule of the shader,
 * src/shaders/SHA-256.comp:337, :363.  In place.
 */
ORACLE_API int oracle_reduce_height(uint32_t *nodes, size_t n, unsigned height, uint32_t out[8])
{
    if (n == 0) return -1;
    for (unsigned lv = 0; lv < height; ++lv) {
        size_t pairs = (n + 1) / 2;
        for (size_t p = 0; p < pairs; ++p) {
            const uint32_t *l = nodes + 16 * p;
            const uint32_t *r = (2 * p + 1 < n) ? l + 8 : l;
            uint32_t h[8];
            oracle_node(l, r, h);
            memcpy(nodes + 8 * p, h, 32);
        }
        n = pairs;
    }
    memcpy(out, nodes, 32);
    return 0;
}

/* Canonical digest bytes of a word-valued node; hash_to_string,
 * src/vkmr/SHA-256plus.cpp:453-469 (printed by print_bytes, src/vkmr/Debug.cpp:38-46). */
ORACLE_API void oracle_words_to_hex(const uint32_t w[8], char hex[65])
{
    static const char d[] = "0123456789abcdef";
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 4; ++b) {
            unsigned v = (w[i] >> (24 - 8 * b)) & 0xffu;
            hex[8 * i + 2 * b] = d[v >> 4];
            hex[8 * i + 2 * b + 1] = d[v & 15];
        }
    hex[64] = 0;
}

/*
 * Leaf digests of a packed batch: data[] words + {start,size} metadata in the
 * layout of Batch::Push (src/vkmr/Batches.cpp:64-121; VkSha256Metadata,
 * src/common/SHA-256defs.h:51-54).  Only `size` bytes of each string are read
 * (SURVEY.md 8a Q3).  `threads` > 1 splits the range over pthreads -- used by the
 * full-size GPU tests so the checker finishes in seconds; the arithmetic per
 * leaf is the same oracle_leaf().
 */
struct span { const uint32_t *data; const uint32_t *meta; uint32_t *out; size_t lo, hi; };

static void *leaf_worker(void *arg)
{
    struct span *s = (struct span *)arg;
    for (size_t i = s->lo; i < s->hi; ++i) {
        uint32_t start = s->meta[2 * i], size = s->meta[2 * i + 1];
        oracle_leaf((const uint8_t *)(s->data + start), size, s->out + 8 * i);
    }
    return NULL;
}

ORACLE_API void oracle_leaves_packed(const uint32_t *data, const uint32_t *meta, size_t n, uint32_t *out, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_t tid[64];
    struct span sp[64];
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        size_t lo = per * (size_t)t, hi = lo + per;
        if (lo >= n) break;
        if (hi > n) hi = n;
        sp[t] = (struct span){data, meta, out, lo, hi};
        if (threads == 1) { leaf_worker(&sp[t]); continue; }
        pthread_create(&tid[t], NULL, leaf_worker, &sp[t]);
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(tid[t], NULL);
}

/* One tree level over [lo,hi) pairs, for the threaded root below. */
struct lvl { const uint32_t *in; uint32_t *out; size_t n, lo, hi; };

static void *level_worker(void *arg)
{
    struct lvl *s = (struct lvl *)arg;
    for (size_t p = s->lo; p < s->hi; ++p) {
        const uint32_t *l = s->in + 16 * p;
        const uint32_t *r = (2 * p + 1 < s->n) ? l + 8 : l;
        oracle_node(l, r, s->out + 8 * p);
    }
    return NULL;
}

/* Same tree as oracle_root_inplace, levels split over pthreads (ping-pong buffer). */
ORACLE_API int oracle_root_mt(uint32_t *nodes, size_t n, uint32_t out[8], int threads)
{
    if (n == 0) return -1;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    uint32_t *tmp = (uint32_t *)malloc(((n + 1) / 2) * 32);
    if (!tmp) return -2;
    uint32_t *in = nodes, *o = tmp;
    do {
        size_t pairs = (n + 1) / 2;
        int use = (pairs < 4096) ? 1 : threads;
        pthread_t tid[64];
        struct lvl sp[64];
        size_t per = (pairs + (size_t)use - 1) / (size_t)use;
        int started = 0;
        for (int t = 0; t < use; ++t) {
            size_t lo = per * (size_t)t, hi = lo + per;
            if (lo >= pairs) break;
            if (hi > pairs) hi = pairs;
            sp[t] = (struct lvl){in, o, n, lo, hi};
            if (use == 1) { level_worker(&sp[t]); continue; }
            pthread_create(&tid[t], NULL, level_worker, &sp[t]);
            ++started;
        }
        for (int t = 0; t < started; ++t) pthread_join(tid[t], NULL);
        n = pairs;
        uint32_t *sw = in; in = o; o = sw;
    } while (n > 1);
    memcpy(out, in, 32);
    free(tmp);
    return 0;
}

/*
 * Root of a newline-separated stream held in memory, with the reference's line
 * rules: '\n' or end of input ends a line, '\r' is kept, empty lines are never
 * leaves (Input::Get, src/vkmr/Inputs.cpp:75-101; run(), src/vkmr/Vkmr.cpp:38-51).
 * Writes the 64-char hex root; *count/*bytes as printed by Vkmr.cpp:55.
 * Returns 0, or -1 when the stream holds no leaf (nothing printed, Vkmr.cpp:52).
 */
ORACLE_API int oracle_root_of_stream(const uint8_t *buf, size_t len, char hex[65], uint64_t *count, uint64_t *bytes)
{
    size_t cap = 1024, n = 0;
    uint64_t total = 0;
    uint32_t *leaves = (uint32_t *)malloc(cap * 32);
    size_t pos = 0;
    while (pos <= len) {
        const uint8_t *nl = (pos < len) ? memchr(buf + pos, '\n', len - pos) : NULL;
        size_t end = nl ? (size_t)(nl - buf) : len;
        size_t l = end - pos;
        if (l) {
            if (n == cap) { cap *= 2; leaves = (uint32_t *)realloc(leaves, cap * 32); }
            oracle_leaf(buf + pos, l, leaves + 8 * n);
            ++n;
            total += l;
        }
        if (!nl) break;
        pos = end + 1;
    }
    if (count) *count = n;
    if (bytes) *bytes = total;
    if (n == 0) { free(leaves); hex[0] = 0; return -1; }
    uint32_t root[8];
    oracle_root_inplace(leaves, n, root);
    oracle_words_to_hex(root, hex);
    free(leaves);
    return 0;
}
