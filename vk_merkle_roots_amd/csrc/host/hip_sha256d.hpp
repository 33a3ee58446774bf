// hip_sha256d.hpp -- the GPU backend: HIP devices by name, and the stream processor.
//
// Takes the place of the reference's vkmr::VkSha256D / VkSha256D::Instance
// (src/vkmr/SHA-256vk.h:31-85): `HipSha256D` enumerates devices and hands out one
// `Instance` per name ("hip:0", "hip:1", ..., and "hip:all" to shard slices across
// every GPU of the node); an Instance is an ISha256D whose Add() streams strings
// into batches -> mappings -> slices -> reductions, all asynchronous, and whose
// Root() drains the pipeline and combines the slice roots.
#pragma once
#include <memory>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "isha256d.hpp"
#include "ops.hpp"

namespace vkmr {

struct HipConfig {
    uint32_t slice_log2 = 0;         // digests per slice as a power of two; 0 = chosen from the devices when the first strings arrive
                                     // (Instance::ChooseSliceLog2, the counterpart of Slices<T>::SliceSize, src/vkmr/Slices.h:421-454):
                                     // the reference's 2^23 = 256 MiB (SHA-256vk.cpp:23), or, with several GPUs and an input file of
                                     // known size, one slice per GPU; never more than the devices' free memory holds.  VKMR_SLICE_LOG2
                                     // sets it; a value that does not fit is clamped with a message
    bool slice_log2_given = false;   // slice_log2 came from the caller (VKMR_SLICE_LOG2), not from the default
    uint64_t expected_input_bytes = 0;  // bytes stdin will deliver when that is known (a regular file), else 0
    uint64_t expected_leaves = 0;       // strings the caller will stage before anything is mapped (StagePacked): every slice of them is resident at once
    size_t batch_bytes = 40u << 20;  // data bytes per batch.  The reference prefers 256 MiB (MegaX, SHA-256vk.cpp:23,
                                     // :247-248).  Here a batch holds one 32 MiB span of stdin (vkmr_main) with room for
                                     // the padding of short lines; pinning it (50 MiB with its metadata) is 8 ms, and
                                     // all of the pipeline's batches are pinned before the first string is read
                                     // (profiles/r03_frontend_phases.txt)
    size_t batch_bytes_max = 1u << 30;  // long strings: batches grow until they hold about 2^19 strings, up to this many
                                     // bytes (a 64 MiB batch of 2 KiB strings is 32 k strings = 512 wavefronts for
                                     // 1024 SIMDs; the map kernel needs >= ~0.5 M strings per launch to fill the chip)
    size_t max_inflight = 3;         // mappings in flight before Add() blocks on the oldest (the pipeline is PCIe-bound: one
                                     // batch being packed, one being copied, one in the kernel, one to spare)
    size_t slice_budget = 0;         // slices resident per device at most (0 = max_inflight + 1: the one being filled
                                     // plus one per mapping in flight); when used up, Add() blocks on the oldest
                                     // reduction and re-uses its slice instead of allocating another
    unsigned pack_threads = 0;       // threads packing large input spans (0 = min(16, hardware threads): 0.23 s with 8,
                                     // 0.20 s with 16 on 2^25 strings, profiles/r02_end_to_end_stdin.txt)
    bool send_sizes = true;          // batches of strings shorter than 65 536 bytes cross PCIe as data + 2 bytes per string; the {start, size}
                                     // entries are written on the device (VKMR_SEND_METADATA=1 sends the 8-byte entries instead, as the reference does)
    int pack_stream = -1;            // the packer's stores: -1 tuned at run time (ordinary vs streaming, PackTuner), 0 / 1 forced (VKMR_PACK_STREAM)
    bool device_split = false;       // VKMR_DEVICE_SPLIT=1: large spans of text cross PCIe as they are and are split into strings on the device
                                     // (vkmr_hip_split_text_async); the host copies them into pinned memory and counts their lines, nothing else.
                                     // Spans that do not qualify (short, or more strings than the slice has room for) take the host packer
    std::vector<unsigned long long> proof_indices;   // also produce the Merkle proofs of these leaves (0-based, stream order; at most 16; README.md:118-120)
    bool verbose = false;            // per-op log lines like the reference prints
    static HipConfig FromEnv();      // VKMR_SLICE_LOG2, VKMR_BATCH_MB / VKMR_BATCH_BYTES, VKMR_BATCH_MAX_MB, VKMR_MAX_INFLIGHT,
                                     // VKMR_SLICE_BUDGET, VKMR_PACK_THREADS, VKMR_SEND_METADATA, VKMR_PACK_STREAM, VKMR_DEVICE_SPLIT, VKMR_PROOF_INDEX, VKMR_VERBOSE
};

class HipSha256D {
public:
    class Instance;

    HipSha256D() = default;          // no HIP call until a device is asked for: `vkmr CPU` never touches the GPU runtime
    explicit operator bool() const { return Count() > 0; }
    bool Has(const ISha256D::name_type&) const;
    // One use per name, like the reference's Get (SHA-256vk.cpp:224-229).
    std::unique_ptr<Instance> Get(const ISha256D::name_type&, const HipConfig& cfg = HipConfig::FromEnv());
    std::vector<ISha256D::name_type> Available() const;
    // " (<marketing name>)" for a "hip:<n>" entry -- the name the reference would have listed -- else "".
    std::string Describe(const ISha256D::name_type&) const;

private:
    int IndexOf(const ISha256D::name_type&) const;   // device index of "hip:<n>" or of a marketing name; -1 if none
    int Count() const;               // enumerates the devices on first use (hipGetDeviceCount: ~50 ms)
    mutable int m_count = -1;
};

class HipSha256D::Instance : public ISha256D {
public:
    Instance(const std::string& name, std::vector<int> devices, const HipConfig& cfg);
    ~Instance() override;

    out_type Root() override;
    bool Add(const arg_type& arg) override { return Add(arg.data(), arg.size()); }
    bool Add(const char* bytes, size_t size) override;
    bool AddLines(const char* buf, size_t len, bool final, Tally* tally) override;
    bool Reset() override { return false; }   // reference IVkSha256DInstance::Reset, SHA-256vk.h:28

    // Pre-packed input (bindings; bench.py's PCIe-inclusive measurement).  StagePacked copies `count` strings that are
    // already in the packed layout into pinned batches, in stream order, with the same batch / slice hand-offs as Add() --
    // but holds the batches back instead of mapping them.  RootOfStaged() then runs exactly what Add() and Root() would
    // have run on them: every batch through Mappings::Map (H2D on the copy stream, map kernel behind it), slices to
    // Reductions as they fill, the combine -- so that "pinned host memory -> root" can be timed on the product's own
    // schedule.  The pools must be large enough to hold the staged batches and slices (cfg.max_inflight, cfg.slice_budget).
    bool StagePacked(const uint32_t* data, const vkmr_metadata* meta, size_t count);
    out_type RootOfStaged();

    bool Ok() const { return m_ok; }
    // "proof: ..." lines of the requested Merkle proof (cfg.proof_indices), valid after Root().
    std::vector<std::string> ProofLines() const { return m_reductions ? m_reductions->ProofLines() : std::vector<std::string>(); }

private:
    struct PerDevice {
        int dev;
        vkmr_stream map_stream = nullptr, copy_stream = nullptr, reduce_stream = nullptr;   // reduce_stream is the map stream (see the constructor)
        std::unique_ptr<Batches> batches;
        bool prefetched = false;   // the pipeline's batches have been requested from the helper thread
    };
    PerDevice& Dev(int dev);
    // The device-split path of AddLines: text[0, len) (whole lines; `final`: the last may lack its newline) into the current,
    // empty, batch as raw text.  Returns the bytes taken (0: this span goes through the host packer instead).
#ifdef VKMR_EXPERIMENTS
    size_t PushTextForDevice(const char* text, size_t len, bool final, Tally* tally);   // device-split path: experiments build only
#endif
    void StartPrefetch();                                // the first device's batches start being pinned on its pool's helper thread
    bool EnsureGeometry(const char* first_span, size_t len);   // slices and reductions exist from the first string on
    uint32_t ChooseSliceLog2(const char* first_span, size_t len, std::string* why) const;
    bool MapCurrent();                                   // dispatches m_batch into the current slice's pending reservations
    void Account(std::vector<Slice>&& retired);          // retired sub-slices -> fill counts -> reductions
    bool StartSliceAndBatch();
    bool NewBatch(int dev);                              // m_batch <- a batch of `dev`, waiting for one to retire if need be
    bool WaitForMemory();                                // blocks on the oldest reduction / mapping; false when nothing is in flight
    void AdaptBatchSize(const Batch& sent);              // long strings: larger batches from now on
    bool GrowBatchesFor(int dev, size_t string_bytes);   // a string larger than the current batches: grow them, if allowed

    HipConfig m_cfg;
    std::vector<int> m_device_ids;
    bool m_geometry = false;
    std::vector<PerDevice> m_devs;
    Slices m_slices;
    Batch m_batch;
    std::unique_ptr<Mappings> m_mappings;
    std::unique_ptr<Reductions> m_reductions;
    std::unique_ptr<class ForkJoin> m_pool;   // packs large input spans in parallel
    std::mutex m_setup_mu;                    // several devices are set up side by side: the first error is kept
    bool m_setup_ok = true;
    std::string m_setup_error;
    bool m_ok;
    bool m_draining = false;   // Root() has begun: nothing more will be packed
    size_t m_free_at_start = ~(size_t)0;   // least free memory over the devices when the instance was made (~0: no device said)
    double m_words_per_byte = 0.0;   // packed words per input byte of the spans packed so far (how much of a span a batch will hold)
    struct Staged { Batch batch; Slice sub; int dev; };
    std::vector<Staged> m_staged;
};

}  // namespace vkmr
