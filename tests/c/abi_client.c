/* abi_client.c -- a plain C99 client of include/vkmr_hip.h, compiled with gcc and linked against
 * libvkmr_hip.so by tests/test_abi_c_client.py: packs its argv strings the way Batch::Push does,
 * maps, reduces, reads the root back and prints it.  Shows that the boundary needs nothing but
 * the header and the library (no C++, no Python, no torch). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vkmr_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        vkmr_status st_ = (call);                                                \
        if (st_ < 0) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, st_, vkmr_hip_last_error()); \
            return 2;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char** argv)
{
    int ndev = 0;
    CHECK(vkmr_hip_device_count(&ndev));
    if (ndev == 0) { fprintf(stderr, "no HIP device\n"); return 3; }
    const uint32_t count = (uint32_t)(argc - 1);
    if (count == 0) return 1;

    /* pack: each string starts on the next word boundary */
    size_t words = 0;
    for (int i = 1; i < argc; ++i) words += (strlen(argv[i]) + 3) / 4;
    void *h_data = NULL, *h_meta = NULL, *h_root = NULL;
    CHECK(vkmr_hip_host_alloc((words ? words : 1) * 4, &h_data));      /* pinned, zero-filled */
    CHECK(vkmr_hip_host_alloc(count * sizeof(vkmr_metadata), &h_meta));
    CHECK(vkmr_hip_host_alloc(sizeof(vkmr_digest), &h_root));
    vkmr_metadata* meta = (vkmr_metadata*)h_meta;
    size_t w = 0;
    for (uint32_t i = 0; i < count; ++i) {
        const size_t len = strlen(argv[i + 1]);
        meta[i].start = (uint32_t)w;
        meta[i].size = (uint32_t)len;
        memcpy((uint32_t*)h_data + w, argv[i + 1], len);
        w += (len + 3) / 4;
    }

    uint32_t height = 1;                       /* a lone leaf is hashed with itself */
    while (((count + (1u << height) - 1) >> height) > 1) ++height;

    vkmr_stream s = NULL;
    vkmr_event done = NULL;
    void *d_data = NULL, *d_meta = NULL, *d_slice = NULL, *d_scratch = NULL, *d_root = NULL;
    CHECK(vkmr_hip_stream_create(0, &s));
    CHECK(vkmr_hip_warm_up(0, s, VKMR_WARM_KERNELS | VKMR_WARM_COPY, (size_t)1 << 20));   /* the reference's pipeline creation at start-up */
    CHECK(vkmr_hip_event_create(0, &done));
    CHECK(vkmr_hip_device_alloc(0, (words ? words : 1) * 4, &d_data));
    CHECK(vkmr_hip_device_alloc(0, count * sizeof(vkmr_metadata), &d_meta));
    CHECK(vkmr_hip_device_alloc(0, count * sizeof(vkmr_digest), &d_slice));
    CHECK(vkmr_hip_device_alloc(0, vkmr_hip_reduce_scratch_bytes(count), &d_scratch));
    CHECK(vkmr_hip_device_alloc(0, sizeof(vkmr_digest), &d_root));

    if (words) CHECK(vkmr_hip_memcpy_h2d_async(0, s, d_data, h_data, words * 4));
    CHECK(vkmr_hip_memcpy_h2d_async(0, s, d_meta, h_meta, count * sizeof(vkmr_metadata)));
    CHECK(vkmr_hip_map_async(0, s, (const uint32_t*)d_data, words, (const vkmr_metadata*)d_meta, count, (vkmr_digest*)d_slice));
    CHECK(vkmr_hip_reduce_async(0, s, (const vkmr_digest*)d_slice, count, height, d_scratch, (vkmr_digest*)d_root));
    CHECK(vkmr_hip_memcpy_d2h_async(0, s, h_root, d_root, sizeof(vkmr_digest)));
    CHECK(vkmr_hip_event_record(0, done, s));
    while (vkmr_hip_event_query(0, done) == VKMR_NOT_READY) { /* the reference polls its fences the same way */ }
    CHECK(vkmr_hip_event_wait(0, done));

    char hex[65];
    vkmr_hip_digest_hex((const vkmr_digest*)h_root, hex);
    printf("%s\n", hex);

    vkmr_hip_device_free(0, d_data); vkmr_hip_device_free(0, d_meta); vkmr_hip_device_free(0, d_slice);
    vkmr_hip_device_free(0, d_scratch); vkmr_hip_device_free(0, d_root);
    vkmr_hip_host_free(h_data); vkmr_hip_host_free(h_meta); vkmr_hip_host_free(h_root);
    vkmr_hip_event_destroy(0, done);
    vkmr_hip_stream_destroy(0, s);
    return 0;
}
