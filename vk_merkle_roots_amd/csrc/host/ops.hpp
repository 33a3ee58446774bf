// ops.hpp -- asynchronous on-device operations: mappings and reductions.
//
// Same two interfaces as the reference's vkmr::Mappings / vkmr::Reductions
// (src/vkmr/Ops.h:23-57), with HIP handles where Vulkan ones were: a stream replaces
// the VkQueue, an event the VkFence, an event pair the timestamp queries.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "batches.hpp"
#include "isha256d.hpp"
#include "slices.hpp"

namespace vkmr {

// Status of the last failed ABI call, in the role VkResult has in the reference.
typedef vkmr_status HipResult;

// Maps input batches into (sub-)slices of device memory.
class Mappings {
public:
    typedef Slice slice_type;
    virtual ~Mappings() = default;

    // Uploads the batch (on `copy_stream`: its two H2D copies run beside the previous batch's map kernel) and hashes
    // it into the sub-slice (on `map_stream`, after an event says the copies have landed), asynchronously.  The batch
    // -- pinned buffers and its own HBM landing zone -- and the sub-slice view are held until the mapping retires.
    // The two streams take the place of the reference's round-robin compute queues (src/vkmr/Devices.cpp:525-538) and
    // of its one submit per batch (src/vkmr/Mappings.cpp:135-232).  copy_stream may equal map_stream.
    virtual HipResult Map(Batch&&, slice_type&&, vkmr_stream map_stream, vkmr_stream copy_stream) = 0;
    // Polls in-flight mappings; returns the sub-slices of those that finished.
    virtual std::vector<slice_type> Update() = 0;
    // Blocks until every in-flight mapping has finished; returns their sub-slices.
    virtual std::vector<slice_type> WaitFor() = 0;
    // Blocks until at most `limit` mappings are in flight (back-pressure; the
    // reference's first to-do, README.md:113); returns the retired sub-slices.
    virtual std::vector<slice_type> WaitUntilAtMost(size_t limit) = 0;
    virtual size_t InFlight() const = 0;
    // true once a mapping has failed on the device (its sub-slice is never reported back, so the
    // run cannot produce a root)
    virtual bool Failed() const = 0;

    // send_sizes: a batch whose strings are all shorter than 65 536 bytes goes to the device as data + 16-bit sizes, the
    // metadata entries are written there (vkmr_hip_metadata_from_sizes_async); otherwise data + entries, as the reference sends them
    static std::unique_ptr<Mappings> New(bool verbose, bool send_sizes = true);
};

// Reduces slices of device memory to their sub-tree roots and combines the roots.
//
// Slice roots stay in HBM, one array per device (slice k, dealt to device (k-1) % D, is entry
// (k-1) / D of its device's array); a pinned host mirror receives each root for the log lines.
// Scratch buffers and events are pooled per device, so a long stream reduces slice after slice
// without allocating.  The final combine runs on the first device: with one device directly over
// its root array, with several after ONE RCCL all-gather of the arrays (vkmr_hip_gather_roots_async)
// -- the step the reference does by reading every root back and hashing on the CPU
// (src/vkmr/Reductions.cpp:56-69, :703-712).
class Reductions {
public:
    typedef Slice slice_type;
    virtual ~Reductions() = default;

    // Starts the reduction of a slice through `height` levels; the slice's memory goes back to
    // its pool when the reduction retires.  When the device has no memory for another scratch
    // buffer, blocks on the oldest reduction in flight and re-uses its buffers.
    virtual HipResult Reduce(slice_type&&, uint32_t height, vkmr_stream) = 0;
    virtual void Update() = 0;
    // Blocks until the oldest reduction in flight has retired; false when none is in flight.
    virtual bool WaitOne() = 0;
    virtual size_t InFlight() const = 0;
    virtual size_t Allocations() const = 0;   // scratch buffers allocated so far (pool bookkeeping, for the log)
    virtual bool Ok() const = 0;              // false once a reduction could not be prepared, dispatched or completed
    // Waits for every reduction, combines the slice roots in slice order and returns
    // the hex root ("" on failure or when nothing was reduced).
    virtual ISha256D::out_type WaitFor() = 0;

    // Merkle proofs (the reference's to-do, README.md:118-120): asks for the authentication path of leaf `leaf_index`
    // (0-based, stream order); up to 16 leaves per run.  The siblings inside a leaf's slice are written BY the reduction of
    // that slice as it hashes them (vkmr_hip_reduce_proofs_async), those above it by the combine of the slice roots at
    // WaitFor().  ProofLines() then gives, per leaf in request order, "proof: ..." lines, bottom level first.
    virtual void RequestProof(uint64_t leaf_index) = 0;
    virtual std::vector<std::string> ProofLines() const = 0;

    // devices: every device slices are dealt to, in dealing order; capacity: digests per slice.
    static std::unique_ptr<Reductions> New(std::vector<int> devices, size_t capacity, bool verbose);
};

}  // namespace vkmr
