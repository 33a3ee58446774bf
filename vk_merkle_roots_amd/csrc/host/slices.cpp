// slices.cpp -- see slices.hpp.
#include "slices.hpp"

#include <iostream>

namespace vkmr {

Slice::Slice(int dev, number_type number, size_type capacity) : m_dev(dev), m_capacity(capacity), m_number(number)
{
    void* p = nullptr;
    std::cout << "Looking for " << capacity * sizeof(vkmr_digest) << " bytes of sliced memory.." << std::endl;
    if (vkmr_hip_device_alloc(dev, capacity * sizeof(vkmr_digest), &p) == VKMR_OK) {
        m_cells = static_cast<vkmr_digest*>(p);
        m_owns = true;
    } else {
        std::cerr << "Failed to allocate slice " << number << ": " << vkmr_hip_last_error() << std::endl;
        m_capacity = 0;
    }
}

Slice::Slice(Slice&& o) noexcept { *this = std::move(o); }

Slice& Slice::operator=(Slice&& o) noexcept
{
    if (this != &o) {
        Release();
        m_dev = o.m_dev; m_cells = o.m_cells; m_owns = o.m_owns;
        m_capacity = o.m_capacity; m_sliced = o.m_sliced; m_reserved = o.m_reserved; m_filled = o.m_filled;
        m_number = o.m_number;
        o.m_cells = nullptr; o.m_owns = false;
        o.m_capacity = o.m_sliced = o.m_reserved = o.m_filled = 0; o.m_number = 0;
    }
    return *this;
}

void Slice::Release()
{
    if (m_owns && m_cells) {
        vkmr_hip_device_free(m_dev, m_cells);
        std::cout << "Deallocated memory for slice " << m_number << ".." << std::endl;
    }
    m_cells = nullptr;
    m_owns = false;
}

Slice Slice::Sub()
{
    Slice view;
    if (m_reserved > 0 && m_cells) {
        view.m_dev = m_dev;
        view.m_cells = m_cells + m_sliced;
        view.m_owns = false;
        view.m_capacity = m_reserved;
        view.m_reserved = m_reserved;
        view.m_number = m_number;
        m_sliced += m_reserved;
        m_reserved = 0;
    }
    return view;
}

Slices::Slices(std::vector<int> devices, size_t capacity) : m_devices(std::move(devices)), m_capacity(capacity) {}

Slice& Slices::operator[](index_type i)
{
    auto it = m_map.find(i);
    return it == m_map.end() ? m_empty : it->second;
}

Slice Slices::Remove(index_type i)
{
    auto it = m_map.find(i);
    if (it == m_map.end()) return Slice();
    Slice s = std::move(it->second);
    m_map.erase(it);
    return s;
}

Slice& Slices::New()
{
    if (m_capacity == 0 || m_devices.empty()) return m_empty;
    const index_type number = m_current + 1;
    Slice s(m_devices[(number - 1) % m_devices.size()], number, m_capacity);
    if (!s) return m_empty;
    auto placed = m_map.emplace(number, std::move(s));
    m_current = number;
    return placed.first->second;
}

}  // namespace vkmr
