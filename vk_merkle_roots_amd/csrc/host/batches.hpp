// batches.hpp -- packed input batches.
//
// Invariant: the strings of a batch lie back to back -- string i + 1 starts on the word after string i, the first at
// word 0 -- whichever of the Push forms appended them.  Their sizes are kept a second time as 16-bit numbers (saturated):
// a batch in which every string is shorter than 65 536 bytes can be described to the device by those alone
// (vkmr_hip_metadata_from_sizes_async), 2 bytes per string over PCIe instead of 8.
//
// A Batch is what the reference's vkmr::Batch is (src/vkmr/Batches.h:31-129): strings
// packed back to back on 4-byte boundaries in a data buffer, plus one {start word,
// size bytes} metadata entry per string, in memory the GPU side can consume.  On
// MI355X the host side is pinned memory (allocated through the C ABI) and each batch
// owns an HBM landing zone of the same size: Mappings copies host -> HBM on the
// op's stream and the map kernel reads HBM.  Buffers are recycled through Batches
// instead of being returned to the system after every mapping (the reference frees
// them, src/vkmr/Mappings.cpp:328-329; recycling is its first to-do, README.md:113).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "stream_pack.hpp"
#include "vkmr_hip.h"

namespace vkmr {

class Batches;

class Batch {
    friend class Batches;

public:
    typedef uint32_t number_type;
    typedef size_t size_type;

    Batch() = default;
    Batch(Batch&&) noexcept;
    Batch& operator=(Batch&&) noexcept;
    Batch(const Batch&) = delete;
    Batch& operator=(const Batch&) = delete;
    ~Batch() { Release(); }

    explicit operator bool() const { return m_data != nullptr && m_meta != nullptr; }

    size_type Count() const { return m_count; }          // strings in the batch
    size_type Size() const { return m_bytes; }           // payload bytes
    size_type Words() const { return m_words; }          // packed words used
    bool Empty() const { return !(*this) || m_count == 0; }
    size_type RoomWords() const { return m_cap_words - m_words; }     // packed words still free
    size_type CapacityWords() const { return m_cap_words; }
    size_type CapacityCount() const { return m_cap_count; }
    number_type Number() const { return m_number; }
    int Device() const { return m_dev; }

    // Appends one string / a group of strings; false when it does not fit (the batch
    // is unchanged).  Reference Batch::Push, src/vkmr/Batches.cpp:64-121.
    bool Push(const char* p, size_t n);
    bool Push(const std::vector<std::string>& strings);
    // Packs the non-empty lines of buf[0,len) straight into the pinned buffers, at most
    // `max_strings`; stops early when the batch is full.  Returns what PackLines reports
    // (bytes consumed, strings appended, ...).
    struct PackResult PushLines(const char* buf, size_t len, bool final, size_t max_strings);
    // Parallel form of PushLines for large spans: the span is cut into pool.Width() parts at line
    // ends, the parts are measured in parallel, the leading parts that fit the batch (and
    // `max_strings`) are packed in parallel at their prefix offsets.  Returns what was consumed
    // and appended (possibly nothing: the caller then falls back to PushLines).  Every part
    // but the last of a final span ends in '\n'.  No more of the span is looked at than the batch is likely to hold:
    // `words_per_byte` is the caller's running figure of packed words per input byte (0 = unknown).
    struct PackResult PushLinesParallel(const char* buf, size_t len, bool final, size_t max_strings, class ForkJoin& pool,
                                        double words_per_byte = 0.0);
    // Appends strings that are already in the packed layout (consecutive, canonical: string i + 1 starts on the word after
    // string i): one memcpy of their words, metadata rebased onto this batch.  Takes as many of the `count` strings as fit
    // (and at most `max_strings`); returns how many.
    size_t PushPacked(const uint32_t* data, const vkmr_metadata* meta, size_t count, size_t max_strings);
    // Drops the last `count` strings (reference Batch::Pop, src/vkmr/Batches.cpp:123-125).
    void Pop(size_t count);

    // host (pinned) and device views, for Mappings
    const uint32_t* HostData() const { return m_data; }
    const vkmr_metadata* HostMeta() const { return m_meta; }
    uint32_t* DeviceData() const { return m_ddata; }
    vkmr_metadata* DeviceMeta() const { return m_dmeta; }
    // the 16-bit sizes: usable in place of the metadata when no string reaches 65 536 bytes
    bool SizesSuffice() const { return m_sizes != nullptr && m_longest < 0xFFFFu; }
    const uint16_t* HostSizes() const { return m_sizes; }
    uint16_t* DeviceSizes() const { return m_dsizes; }
    void* DeviceSizesScratch() const { return m_dscratch; }
    // Device-side splitting (pools made with `device_split`): the batch holds raw text instead of packed strings.  The text
    // (whole lines, ending in '\n') is written to TextArea() -- the pinned data buffer taken as bytes -- and SetText says how
    // long it is and what it holds; the device turns it into the packed layout in its own copy of the batch
    // (vkmr_hip_split_text_async) and reports what it found in HostSplitResult() for the owner to check.
    bool CanHoldText() const { return m_dtext != nullptr; }
    uint8_t* TextArea() const { return reinterpret_cast<uint8_t*>(m_data); }
    size_t TextCapacity() const { return m_cap_words * 4; }
    void SetText(size_t text_bytes, size_t strings, size_t payload_bytes);
    size_t TextBytes() const { return m_text_bytes; }
    uint8_t* DeviceText() const { return m_dtext; }
    void* DeviceSplitScratch() const { return m_dsplit; }
    uint32_t* DeviceSplitResult() const { return m_dresult; }
    uint32_t* HostSplitResult() const { return m_hresult; }

private:
    void Release();

    Batches* m_owner = nullptr;
    int m_dev = -1;
    uint32_t* m_data = nullptr;        // pinned host, data_words capacity
    vkmr_metadata* m_meta = nullptr;   // pinned host, meta capacity
    uint32_t* m_ddata = nullptr;       // HBM
    vkmr_metadata* m_dmeta = nullptr;  // HBM
    uint16_t* m_sizes = nullptr;       // pinned host, meta capacity
    uint16_t* m_dsizes = nullptr;      // HBM
    void* m_dscratch = nullptr;        // HBM, vkmr_hip_sizes_scratch_bytes(meta capacity)
    uint8_t* m_dtext = nullptr;        // HBM: the raw text of a device-split batch (pools with device_split only)
    void* m_dsplit = nullptr;          // HBM: vkmr_hip_split_scratch_bytes
    uint32_t* m_dresult = nullptr;     // HBM: the splitter's three result words
    uint32_t* m_hresult = nullptr;     // pinned host: their copy
    size_t m_text_bytes = 0;           // > 0: the batch holds that much raw text, not packed strings
    size_t m_longest = 0;              // longest string appended so far
    void NoteSizes(size_t first, size_t count);   // sizes[first, first + count) <- meta, m_longest
    size_t m_cap_words = 0, m_cap_count = 0;
    size_t m_count = 0, m_words = 0, m_bytes = 0;
    number_type m_number = 0xFFFFFFFFu;
};

// Which form of the packer's second pass a pool's batches use: ordinary stores leave the packed words in the cache, where
// the copy engine finds them on a quiet host; streaming stores save the read-for-ownership of every destination line,
// which is what counts when the host's memory is busy (profiles/r03_frontend_streaming_stores.txt: either can be 20-30 %
// faster than the other).  The tuner times both -- alternately at first, then the slower one every sixteenth call -- and
// hands out the form that has been moving more bytes per second of late.
class PackTuner {
public:
    explicit PackTuner(int forced = -1) : m_forced(forced) {}
    bool Next()   // true: streaming stores for this call
    {
        if (m_forced >= 0) return m_forced != 0;
        const unsigned k = m_calls++;
        if (k < 8) return (k & 1u) != 0;                       // four calls each to begin with
        const bool better = m_rate[1] > m_rate[0];             // bytes per second, smoothed
        return (k % 16u == 15u) ? !better : better;            // the other form now and then: the host's state changes
    }
    void Report(bool streaming, size_t bytes, double seconds)
    {
        if (bytes == 0 || seconds <= 0.0) return;
        const double rate = (double)bytes / seconds;
        double& r = m_rate[streaming ? 1 : 0];
        r = r == 0.0 ? rate : 0.75 * r + 0.25 * rate;
    }
    double Rate(bool streaming) const { return m_rate[streaming ? 1 : 0]; }

private:
    int m_forced;
    unsigned m_calls = 0;
    double m_rate[2] = {0.0, 0.0};
};

// Allocates and recycles batches for one device.
class Batches {
public:
    // data_bytes: capacity of a batch's data buffer; metadata capacity follows the
    // reference's ratio (one entry per 32 data bytes, src/vkmr/Batches.h:131-134).
    // device_split: every batch also gets what vkmr_hip_split_text_async needs (a text area in HBM, scratch, result words)
    // pack_stream: -1 the tuner decides between ordinary and streaming stores in the packer, 0 / 1 forces one
    Batches(int dev, size_t data_bytes, bool device_split = false, int pack_stream = -1);
    PackTuner& Tuner() { return m_tuner; }
    ~Batches();
    Batches(const Batches&) = delete;
    Batches& operator=(const Batches&) = delete;

    // Starts allocating `n` batches of the current shape on a helper thread.  Pinning 80 MiB of host memory takes
    // 10-15 ms; a stream that needs five batches would otherwise spend 60 ms of its first 100 allocating on the
    // thread that should be packing (profiles/r02_hip_api_stats_256_slices.csv).
    void Prefetch(size_t n);
    Batch New();                 // a prefetched, recycled or fresh batch; falsy when allocation fails
    void Recycle(Batch& b);      // called by Batch::Release
    size_t InCirculation() const { return m_live; }
    size_t Allocations() const { return m_allocations; }
    size_t DataBytes() const { return m_words * 4; }
    size_t MetaCount() const { return m_count; }
    // Batches handed out from now on hold `data_bytes` of data and `meta_count` strings; buffers of
    // the old shape are released as they come back.
    void Reshape(size_t data_bytes, size_t meta_count);

private:
    struct Buffers { uint32_t* data; vkmr_metadata* meta; uint32_t* ddata; vkmr_metadata* dmeta; size_t words, count; uint16_t* sizes; uint16_t* dsizes; void* dscratch;
                     uint8_t* dtext; void* dsplit; uint32_t* dresult; uint32_t* hresult; };
    void Free(Buffers& b);
    bool Allocate(size_t words, size_t count, Buffers* out);
    void JoinPrefetch();
    int m_dev;
    bool m_device_split = false;
    PackTuner m_tuner;
    size_t m_words, m_count, m_live, m_allocations = 0;
    uint32_t m_next;
    std::vector<Buffers> m_free;
    // the prefetch thread and the owner share m_free, m_pending and m_allocations under m_mu
    std::mutex m_mu;
    std::condition_variable m_cv;
    std::thread m_prefetch;
    size_t m_pending = 0;        // batches the prefetch thread has still to deliver
    std::atomic<bool> m_stop{false};
};

}  // namespace vkmr
