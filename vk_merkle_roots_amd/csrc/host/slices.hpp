// slices.hpp -- device-resident slices of digests.
//
// A Slice is the reference's vkmr::Slice<VkSha256Result> (src/vkmr/Slices.h:31-251):
// an HBM array of `capacity` digest cells, a power of two, numbered from 1, filled by
// reservations and handed to mappings as sub-slices.  Vulkan needed a VkBuffer per
// sub-slice bound at an aligned offset (src/vkmr/Slices.h:145-187, AlignedReservationSize
// :107-109); with HIP a sub-slice is just a pointer offset, so the aligned
// reservation size is 1.
//
// The HBM of a retired slice is kept and handed to the next slice of the same device
// instead of going back to the driver (the reference frees it, src/vkmr/Slices.h:231-241;
// re-using it is its first to-do, README.md:113): in the steady state of a long stream no
// slice is ever hipMalloc'd or hipFree'd -- each of those is a device-wide synchronisation
// in the middle of an otherwise asynchronous pipeline.
#pragma once
#include <cstdint>
#include <memory>
#include <unordered_map>
#include <utility>
#include <vector>

#include "vkmr_hip.h"

namespace vkmr {

// The slice memory of one run: equal-sized digest arrays per device, a free list of the ones
// whose reduction has retired, and an optional budget of resident slices per device.
class SlicePool {
public:
    SlicePool(size_t capacity_cells, size_t budget_per_device) : m_capacity(capacity_cells), m_budget(budget_per_device) {}
    ~SlicePool();
    SlicePool(const SlicePool&) = delete;
    SlicePool& operator=(const SlicePool&) = delete;

    // A digest array of the pool's capacity on `dev`: a retired one when there is one, otherwise
    // freshly allocated; nullptr when the budget is used up or HBM is (`*budget_hit` tells which).
    vkmr_digest* Acquire(int dev, bool* budget_hit);
    void Release(int dev, vkmr_digest* cells);
    size_t Resident(int dev) const;
    size_t Allocations() const { return m_allocations; }

private:
    struct PerDevice { int dev; size_t resident; std::vector<vkmr_digest*> free; };
    PerDevice& Dev(int dev);
    size_t m_capacity, m_budget, m_allocations = 0;
    std::vector<PerDevice> m_devs;
};

class Slice {
public:
    typedef uint32_t number_type;
    typedef size_t size_type;

    Slice() = default;
    Slice(std::shared_ptr<SlicePool> pool, int dev, number_type number, size_type capacity, bool* budget_hit);   // takes HBM from the pool
    Slice(Slice&&) noexcept;
    Slice& operator=(Slice&&) noexcept;
    Slice(const Slice&) = delete;
    Slice& operator=(const Slice&) = delete;
    ~Slice() { Release(); }

    explicit operator bool() const { return m_cells != nullptr; }

    // a retired mapping reports its sub-slice back (reference operator+=, Slices.h:84-89)
    Slice& operator+=(const Slice& sub)
    {
        if (sub.Number() == Number()) m_filled += sub.Reserved();
        return *this;
    }
    bool IsFilled() const { return m_filled >= m_capacity; }

    number_type Number() const { return m_number; }
    int Device() const { return m_dev; }
    size_type AlignedReservationSize() const { return 1; }
    size_type Available() const { return m_capacity - (m_sliced + m_reserved); }
    bool Reserve(size_type count = 1)
    {
        if (Available() < count) return false;
        m_reserved += count;
        return true;
    }
    void Unreserve(size_type count = 1) { m_reserved -= (count < m_reserved ? count : m_reserved); }
    size_type Reserved() const { return m_reserved; }
    size_type Count() const { return m_sliced; }
    size_type Capacity() const { return m_capacity; }
    size_type Filled() const { return m_filled; }
    vkmr_digest* Cells() const { return m_cells; }

    // The reservations made since the last call, as a non-owning view (reference
    // Slice::Sub, Slices.h:145-187).
    Slice Sub();

private:
    void Release();

    int m_dev = -1;
    vkmr_digest* m_cells = nullptr;
    bool m_owns = false;
    std::shared_ptr<SlicePool> m_pool;   // where an owning slice returns its memory
    size_type m_capacity = 0, m_sliced = 0, m_reserved = 0, m_filled = 0;
    number_type m_number = 0;
};

// The slices of one run, numbered 1, 2, ... in stream order (reference Slices<T>,
// src/vkmr/Slices.h:253-478).  Slice k lives on device devices[(k-1) % devices.size()].
class Slices {
public:
    typedef Slice::number_type index_type;

    Slices() = default;
    // budget: slices resident per device at most (0 = as many as HBM holds)
    Slices(std::vector<int> devices, size_t capacity, size_t budget_per_device = 0);

    Slice& operator[](index_type i);
    Slice& Current() { return (*this)[m_current]; }
    Slice Remove(index_type i);
    // The next slice; falsy when its device has no memory for it right now.  `*budget_hit` (optional)
    // tells a used-up budget from a failed allocation.  The caller waits for a reduction to retire
    // and tries again (Instance::StartSliceAndBatch).
    Slice& New(bool* budget_hit = nullptr);
    size_t Allocations() const { return m_pool ? m_pool->Allocations() : 0; }
    bool Has() const { return !m_map.empty(); }
    const Slice& Any() const { return Has() ? m_map.begin()->second : m_empty; }
    size_t Capacity() const { return m_capacity; }
    index_type LastNumber() const { return m_current; }

private:
    std::vector<int> m_devices;
    std::shared_ptr<SlicePool> m_pool;
    size_t m_capacity = 0;
    index_type m_current = 0;
    std::unordered_map<index_type, Slice> m_map;
    Slice m_empty;
};

}  // namespace vkmr
