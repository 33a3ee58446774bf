// slices.hpp -- device-resident slices of digests.
//
// A Slice is the reference's vkmr::Slice<VkSha256Result> (src/vkmr/Slices.h:31-251):
// an HBM array of `capacity` digest cells, a power of two, numbered from 1, filled by
// reservations and handed to mappings as sub-slices.  Vulkan needed a VkBuffer per
// sub-slice bound at an aligned offset (src/vkmr/Slices.h:145-187, AlignedReservationSize
// :107-109); with HIP a sub-slice is just a pointer offset, so the aligned
// reservation size is 1.
#pragma once
#include <cstdint>
#include <unordered_map>
#include <utility>
#include <vector>

#include "vkmr_hip.h"

namespace vkmr {

class Slice {
public:
    typedef uint32_t number_type;
    typedef size_t size_type;

    Slice() = default;
    Slice(int dev, number_type number, size_type capacity);   // allocates HBM
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
    size_type m_capacity = 0, m_sliced = 0, m_reserved = 0, m_filled = 0;
    number_type m_number = 0;
};

// The slices of one run, numbered 1, 2, ... in stream order (reference Slices<T>,
// src/vkmr/Slices.h:253-478).  Slice k lives on device devices[(k-1) % devices.size()].
class Slices {
public:
    typedef Slice::number_type index_type;

    Slices() = default;
    Slices(std::vector<int> devices, size_t capacity);

    Slice& operator[](index_type i);
    Slice& Current() { return (*this)[m_current]; }
    Slice Remove(index_type i);
    Slice& New();
    bool Has() const { return !m_map.empty(); }
    const Slice& Any() const { return Has() ? m_map.begin()->second : m_empty; }
    size_t Capacity() const { return m_capacity; }
    index_type LastNumber() const { return m_current; }

private:
    std::vector<int> m_devices;
    size_t m_capacity = 0;
    index_type m_current = 0;
    std::unordered_map<index_type, Slice> m_map;
    Slice m_empty;
};

}  // namespace vkmr
