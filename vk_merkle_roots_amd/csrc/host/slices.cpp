// slices.cpp -- see slices.hpp.
#include "slices.hpp"

#include <iostream>

namespace vkmr {

SlicePool::~SlicePool()
{
    for (auto& d : m_devs)
        for (vkmr_digest* p : d.free) vkmr_hip_device_free(d.dev, p);
}

SlicePool::PerDevice& SlicePool::Dev(int dev)
{
    for (auto& d : m_devs)
        if (d.dev == dev) return d;
    m_devs.push_back({dev, 0, {}});
    return m_devs.back();
}

size_t SlicePool::Resident(int dev) const
{
    for (const auto& d : m_devs)
        if (d.dev == dev) return d.resident;
    return 0;
}

vkmr_digest* SlicePool::Acquire(int dev, bool* budget_hit)
{
    PerDevice& d = Dev(dev);
    if (budget_hit) *budget_hit = false;
    if (!d.free.empty()) {
        vkmr_digest* p = d.free.back();
        d.free.pop_back();
        ++d.resident;
        return p;
    }
    if (m_budget && d.resident >= m_budget) {
        if (budget_hit) *budget_hit = true;
        return nullptr;
    }
    void* p = nullptr;
    std::cout << "Looking for " << m_capacity * sizeof(vkmr_digest) << " bytes of sliced memory.." << std::endl;
    if (vkmr_hip_device_alloc(dev, m_capacity * sizeof(vkmr_digest), &p) != VKMR_OK) return nullptr;
    ++m_allocations;
    ++d.resident;
    return static_cast<vkmr_digest*>(p);
}

void SlicePool::Release(int dev, vkmr_digest* cells)
{
    PerDevice& d = Dev(dev);
    d.free.push_back(cells);
    if (d.resident) --d.resident;
}

Slice::Slice(std::shared_ptr<SlicePool> pool, int dev, number_type number, size_type capacity, bool* budget_hit)
    : m_dev(dev), m_capacity(capacity), m_number(number)
{
    m_cells = pool->Acquire(dev, budget_hit);
    if (m_cells) {
        m_owns = true;
        m_pool = std::move(pool);
    } else {
        m_capacity = 0;
    }
}

Slice::Slice(Slice&& o) noexcept { *this = std::move(o); }

Slice& Slice::operator=(Slice&& o) noexcept
{
    if (this != &o) {
        Release();
        m_dev = o.m_dev; m_cells = o.m_cells; m_owns = o.m_owns; m_pool = std::move(o.m_pool);
        m_capacity = o.m_capacity; m_sliced = o.m_sliced; m_reserved = o.m_reserved; m_filled = o.m_filled;
        m_number = o.m_number;
        o.m_cells = nullptr; o.m_owns = false;
        o.m_capacity = o.m_sliced = o.m_reserved = o.m_filled = 0; o.m_number = 0;
    }
    return *this;
}

void Slice::Release()
{
    if (m_owns && m_cells && m_pool) m_pool->Release(m_dev, m_cells);   // kept for the next slice of this device
    m_cells = nullptr;
    m_owns = false;
    m_pool.reset();
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

Slices::Slices(std::vector<int> devices, size_t capacity, size_t budget_per_device)
    : m_devices(std::move(devices)), m_pool(std::make_shared<SlicePool>(capacity, budget_per_device)), m_capacity(capacity)
{
}

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

Slice& Slices::New(bool* budget_hit)
{
    if (budget_hit) *budget_hit = false;
    if (m_capacity == 0 || m_devices.empty() || !m_pool) return m_empty;
    const index_type number = m_current + 1;
    Slice s(m_pool, m_devices[(number - 1) % m_devices.size()], number, m_capacity, budget_hit);
    if (!s) return m_empty;
    auto placed = m_map.emplace(number, std::move(s));
    m_current = number;
    return placed.first->second;
}

}  // namespace vkmr
