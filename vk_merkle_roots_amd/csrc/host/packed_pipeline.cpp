// packed_pipeline.cpp -- libvkmr_pipeline.so: the GPU backend's stream processor behind two C entry points (bindings):
// vkmr_host_pipeline_text takes newline-separated text in memory -- what `vkmr hip:<n>` reads from stdin, the whole of
// the reference's run() (src/vkmr/Vkmr.cpp:28-58) as one call; vkmr_host_pipeline_packed takes input which is ALREADY in
// the packed batch layout (bench.py's PCIe-inclusive measurement).
//
// The strings are copied into the stream processor's own pinned batches first (untimed); the timed part is what
// `vkmr hip:<n>` runs per batch and per slice -- Mappings::Map (H2D copies on the device's copy stream, the map kernel
// behind them on its map stream), Reductions as slices fill, the combine of the slice roots -- so the figure is the
// product's schedule, not a hand-rolled loop.  Reference: one submit per batch
// (src/vkmr/Mappings.cpp:135-232) on round-robin queues (src/vkmr/Devices.cpp:525-538).
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "hip_sha256d.hpp"

extern "C" {

// device: a HIP device index, or -1 for every device ("hip:all").  strings_per_batch: how many strings one map launch
// takes (the batch pool is sized for it); slice_log2: digests per slice, 0 = chosen from the devices.
// root_hex: 65 bytes.  seconds: the timed part (first Map to the root on the host).  Returns 0, or -1 on failure.
__attribute__((visibility("default"))) int vkmr_host_pipeline_packed(int device, const uint32_t* data, uint64_t words, const vkmr_metadata* meta,
                                                                      uint64_t count, uint64_t strings_per_batch, uint32_t slice_log2,
                                                                      char* root_hex, double* seconds)
{
    if (!data || !meta || !root_hex || count == 0 || strings_per_batch == 0) return -1;
    vkmr::HipSha256D gpus;
    const std::string name = device < 0 ? "hip:all" : "hip:" + std::to_string(device);
    if (!gpus.Has(name)) return -1;
    vkmr::HipConfig cfg = vkmr::HipConfig::FromEnv();
    const uint64_t nbatches = (count + strings_per_batch - 1) / strings_per_batch;
    // a batch holds strings_per_batch strings of the average size (+ 2 % and a page), and every batch is staged before
    // the first one is mapped: the pools must hold all of them
    const double avg_words = (double)words / (double)count;
    cfg.batch_bytes = (size_t)((double)strings_per_batch * avg_words * 4.0 * 1.02) + 4096;
    cfg.batch_bytes_max = cfg.batch_bytes;
    cfg.max_inflight = (size_t)nbatches + 1;
    if (slice_log2) { cfg.slice_log2 = slice_log2; cfg.slice_log2_given = true; }
    // every slice is staged before the first is reduced: the budget is what `count` strings fill at the slice size in use --
    // the given one, or whatever the instance chooses from the device's memory (it raises the budget itself then)
    uint32_t sl = slice_log2 ? slice_log2 : 23;
    const uint64_t nslices = (count + ((uint64_t)1 << sl) - 1) >> sl;
    cfg.slice_budget = (size_t)nslices + 1;
    cfg.expected_leaves = count;
    auto inst = gpus.Get(name, cfg);
    if (!inst || !inst->Ok()) return -1;
    // stage, batch by batch (strings_per_batch strings each: the launch shape asked for)
    for (uint64_t at = 0; at < count; at += strings_per_batch) {
        const uint64_t n = (count - at < strings_per_batch) ? count - at : strings_per_batch;
        if (!inst->StagePacked(data, meta + at, (size_t)n)) return -1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const std::string root = inst->RootOfStaged();
    const auto t1 = std::chrono::steady_clock::now();
    if (root.size() != 64) return -1;
    std::memcpy(root_hex, root.c_str(), 65);
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    return 0;
}

// The front end on text in memory: the non-empty lines of text[0,len) -- a line ends at '\n' or at len; '\r' is kept
// (Input::Get, reference src/vkmr/Inputs.cpp:75-101) -- through the parallel packer, pinned batches, Mappings, Reductions and
// the combine, exactly as `vkmr hip:<n> < file` runs them, span by span (span_bytes at a time; 0 = the 32 MiB of vkmr_main).
// device: a HIP device index, or -1 for every device ("hip:all").  root_hex: 65 bytes; all zero-length when the text holds no
// string (the reference prints no root then).  items / bytes: strings added and their payload bytes.  seconds: from the
// first span to the root on the host (the reference's stopwatch).  Returns 0, or -1 on failure.
__attribute__((visibility("default"))) int vkmr_host_pipeline_text(int device, const char* text, uint64_t len, uint64_t span_bytes, char* root_hex,
                                                                    uint64_t* items, uint64_t* bytes, double* seconds)
{
    if ((!text && len != 0) || !root_hex) return -1;
    vkmr::HipSha256D gpus;
    const std::string name = device < 0 ? "hip:all" : "hip:" + std::to_string(device);
    if (!gpus.Has(name)) return -1;
    vkmr::HipConfig cfg = vkmr::HipConfig::FromEnv();
    cfg.expected_input_bytes = len;
    auto inst = gpus.Get(name, cfg);
    if (!inst || !inst->Ok()) return -1;
    const uint64_t span = span_bytes ? span_bytes : ((uint64_t)32 << 20);
    uint64_t n_items = 0, n_bytes = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t at = 0; at < len;) {
        uint64_t take = len - at;
        if (take > span) {   // whole lines only: up to the last '\n' of the span, or -- a line longer than the span -- to its end
            const void* nl = memrchr(text + at, '\n', (size_t)span);
            if (!nl) nl = memchr(text + at + span, '\n', (size_t)(take - span));
            if (nl) take = (uint64_t)(static_cast<const char*>(nl) - (text + at)) + 1;
        }
        vkmr::ISha256D::Tally tally;
        if (!inst->AddLines(text + at, (size_t)take, at + take == len, &tally)) return -1;
        n_items += tally.items;
        n_bytes += tally.bytes;
        at += take;
    }
    root_hex[0] = 0;
    if (n_items > 0) {
        const std::string root = inst->Root();
        if (root.size() != 64) return -1;
        std::memcpy(root_hex, root.c_str(), 65);
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (items) *items = n_items;
    if (bytes) *bytes = n_bytes;
    return 0;
}

}  // extern "C"
