// packed_pipeline_client.cpp -- test driver (tests/test_host_pipeline.py): vkmr_host_pipeline_packed on a packed batch read from two
// files, as a sanitized executable (an ASan library cannot be loaded into the Python test process).
//   packed_pipeline_client <data.u32> <meta.u32x2> <strings per batch> <slice log2> [device]
// Prints "rc=<rc> root=<hex>".
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vkmr_hip.h"

extern "C" int vkmr_host_pipeline_packed(int device, const uint32_t* data, uint64_t words, const vkmr_metadata* meta, uint64_t count,
                                         uint64_t strings_per_batch, uint32_t slice_log2, char* root_hex, double* seconds);

template <typename T>
static std::vector<T> slurp(const char* path)
{
    std::vector<T> v;
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / sizeof(T));
    if (n > 0 && fread(v.data(), 1, (size_t)n, f) != (size_t)n) { perror("fread"); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const std::vector<uint32_t> data = slurp<uint32_t>(argv[1]);
    const std::vector<vkmr_metadata> meta = slurp<vkmr_metadata>(argv[2]);
    char hex[65] = "";
    double secs = 0;
    const int rc = vkmr_host_pipeline_packed(argc > 5 ? atoi(argv[5]) : 0, data.data(), data.size(), meta.data(), meta.size(),
                                             (uint64_t)atoll(argv[3]), (uint32_t)atoi(argv[4]), hex, &secs);
    printf("rc=%d root=%s\n", rc, rc == 0 ? hex : "");
    return 0;
}
