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

    // Uploads the batch and hashes it into the sub-slice, asynchronously.  The batch
    // and the sub-slice view are held until the mapping retires.
    virtual HipResult Map(Batch&&, slice_type&&, vkmr_stream) = 0;
    // Polls in-flight mappings; returns the sub-slices of those that finished.
    virtual std::vector<slice_type> Update() = 0;
    // Blocks until every in-flight mapping has finished; returns their sub-slices.
    virtual std::vector<slice_type> WaitFor() = 0;
    // Blocks until at most `limit` mappings are in flight (back-pressure; the
    // reference's first to-do, README.md:113); returns the retired sub-slices.
    virtual std::vector<slice_type> WaitUntilAtMost(size_t limit) = 0;
    virtual size_t InFlight() const = 0;

    static std::unique_ptr<Mappings> New(bool verbose);
};

// Reduces slices of device memory to their sub-tree roots and combines the roots.
class Reductions {
public:
    typedef Slice slice_type;
    virtual ~Reductions() = default;

    // Starts the reduction of a slice through `height` levels; the slice's memory is
    // released when the reduction retires.
    virtual HipResult Reduce(slice_type&&, uint32_t height, vkmr_stream) = 0;
    virtual void Update() = 0;
    // Waits for every reduction, combines the slice roots in slice order and returns
    // the hex root ("" on failure or when nothing was reduced).
    virtual ISha256D::out_type WaitFor() = 0;

    static std::unique_ptr<Reductions> New(int combine_device, bool verbose);
};

}  // namespace vkmr
