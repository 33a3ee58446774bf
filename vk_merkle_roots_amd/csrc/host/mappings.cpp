// mappings.cpp -- in-flight mappings (reference src/vkmr/Mappings.cpp:294-365, without
// the descriptor/command-buffer machinery: one mapping = two async copies on the device's copy
// stream, an event, one kernel launch on its map stream, and the events that time and retire it).
#include <iostream>

#include "ops.hpp"
#ifdef VKMR_EXPERIMENTS
#include "vkmr_hip_experiments.h"
#endif
#include "timing.hpp"

namespace vkmr {
namespace {

struct Mapping {
    Batch batch;
    Slice sub;
    vkmr_event begin = nullptr, copied = nullptr, done = nullptr;
    int dev = -1;
};

class MappingsImpl : public Mappings {
public:
    MappingsImpl(bool verbose, bool send_sizes) : m_verbose(verbose), m_send_sizes(send_sizes) {}
    ~MappingsImpl() override
    {
        WaitFor();
        for (auto& e : m_spare) vkmr_hip_event_destroy(e.first, e.second);
    }

    HipResult Map(Batch&& batch, slice_type&& sub, vkmr_stream stream, vkmr_stream copy_stream) override
    {
        if (batch.Empty() || !sub) return VKMR_OK;   // nothing to do
        Mapping m;
        m.dev = sub.Device();
        m.begin = Event(m.dev);
        m.copied = Event(m.dev);
        m.done = Event(m.dev);
        auto give_back = [&] {
            for (vkmr_event e : {m.begin, m.copied, m.done})
                if (e) m_spare.emplace_back(m.dev, e);   // keep the events for the next mapping
        };
        if (!m.begin || !m.copied || !m.done) {
            give_back();
            return VKMR_ERR_HIP;
        }
        HipResult r = vkmr_hip_event_record(m.dev, m.begin, copy_stream);
#ifdef VKMR_EXPERIMENTS
        if (batch.TextBytes() > 0) {
            // (experiments build) raw text: it crosses as it is, the device splits it into the packed layout in the batch's
            // landing zone and says what it found (checked when the mapping retires)
            timing::Scope ts(timing::MAP_COPIES);
            if (r == VKMR_OK) r = vkmr_hip_memcpy_h2d_async(m.dev, copy_stream, batch.DeviceText(), batch.TextArea(), batch.TextBytes());
            if (r == VKMR_OK)
                r = vkmr_hip_split_text_async(m.dev, copy_stream, batch.DeviceText(), (uint32_t)batch.TextBytes(), batch.DeviceSplitScratch(), batch.DeviceData(),
                                              batch.CapacityWords(), batch.DeviceMeta(), (uint32_t)batch.CapacityCount(), batch.DeviceSplitResult());
            if (r == VKMR_OK) r = vkmr_hip_memcpy_d2h_async(m.dev, copy_stream, batch.HostSplitResult(), batch.DeviceSplitResult(), 3 * sizeof(uint32_t));
        } else
#endif
        {
            timing::Scope ts(timing::MAP_COPIES);
            if (r == VKMR_OK)
                r = vkmr_hip_memcpy_h2d_async(m.dev, copy_stream, batch.DeviceData(), batch.HostData(), batch.Words() * 4);
            if (r == VKMR_OK && m_send_sizes && batch.SizesSuffice()) {
                // 2 bytes per string over the link instead of 8: the entries are written on the device from the sizes
                // (the strings of a batch lie back to back from word 0: batches.hpp)
                r = vkmr_hip_memcpy_h2d_async(m.dev, copy_stream, batch.DeviceSizes(), batch.HostSizes(), batch.Count() * sizeof(uint16_t));
                if (r == VKMR_OK)
                    r = vkmr_hip_metadata_from_sizes_async(m.dev, copy_stream, batch.DeviceSizes(), (uint32_t)batch.Count(), 0u,
                                                           batch.DeviceSizesScratch(), batch.DeviceMeta());
            } else if (r == VKMR_OK) {
                r = vkmr_hip_memcpy_h2d_async(m.dev, copy_stream, batch.DeviceMeta(), batch.HostMeta(),
                                              batch.Count() * sizeof(vkmr_metadata));
            }
        }
        if (r == VKMR_OK && copy_stream != stream) {
            r = vkmr_hip_event_record(m.dev, m.copied, copy_stream);
            if (r == VKMR_OK) r = vkmr_hip_stream_wait_event(m.dev, stream, m.copied);
        }
        if (r == VKMR_OK) {
            timing::Scope ts(timing::MAP_LAUNCH);
            r = vkmr_hip_map_async(m.dev, stream, batch.DeviceData(), batch.Words(), batch.DeviceMeta(),
                                   (uint32_t)batch.Count(), sub.Cells());
        }
        if (r == VKMR_OK) r = vkmr_hip_event_record(m.dev, m.done, stream);
        if (r != VKMR_OK) {
            std::cerr << "Failed to dispatch a mapping: " << vkmr_hip_last_error() << std::endl;
            give_back();
            return r;
        }
        m.batch = std::move(batch);
        m.sub = std::move(sub);
        m_inflight.push_back(std::move(m));
        return VKMR_OK;
    }

    std::vector<slice_type> Update() override { return Retire(false, (size_t)-1); }
    std::vector<slice_type> WaitFor() override { return Retire(true, 0); }
    std::vector<slice_type> WaitUntilAtMost(size_t limit) override { return Retire(true, limit); }
    size_t InFlight() const override { return m_inflight.size(); }
    bool Failed() const override { return m_failed; }

private:
    vkmr_event Event(int dev)
    {
        for (size_t i = 0; i < m_spare.size(); ++i)
            if (m_spare[i].first == dev) {
                vkmr_event e = m_spare[i].second;
                m_spare.erase(m_spare.begin() + i);
                return e;
            }
        vkmr_event e = nullptr;
        if (vkmr_hip_event_create(dev, &e) != VKMR_OK) return nullptr;
        return e;
    }

    // Retires finished mappings, oldest first.  With `block`, waits for the oldest
    // until no more than `keep` remain.
    std::vector<slice_type> Retire(bool block, size_t keep)
    {
        std::vector<slice_type> out;
        for (auto it = m_inflight.begin(); it != m_inflight.end();) {
            HipResult st = vkmr_hip_event_query(it->dev, it->done);
            if (st == VKMR_NOT_READY && block && m_inflight.size() > keep) st = vkmr_hip_event_wait(it->dev, it->done);
            if (st == VKMR_NOT_READY) {
                ++it;
                continue;
            }
            if (st < 0) {   // the device reported an error: these digests do not exist
                std::cerr << "Mapping for slice #" << it->sub.Number() << " failed: " << vkmr_hip_last_error() << std::endl;
                m_failed = true;
                m_spare.emplace_back(it->dev, it->begin);
                m_spare.emplace_back(it->dev, it->copied);
                m_spare.emplace_back(it->dev, it->done);
                it = m_inflight.erase(it);
                continue;
            }
            if (it->batch.TextBytes() > 0) {   // the device's splitter and the host's count must agree, and the strings must have fitted
                const uint32_t* found = it->batch.HostSplitResult();
                if (found[0] != it->batch.Count() || found[2] != 0u || found[1] > it->batch.Words()) {
                    std::cerr << "Mapping for slice #" << it->sub.Number() << " failed: the device split the text into " << found[0] << " string(s) in " << found[1]
                              << " word(s)" << (found[2] ? ", more than the batch holds" : "") << "; the host counted " << it->batch.Count() << " in at most "
                              << it->batch.Words() << "." << std::endl;
                    m_failed = true;
                    m_spare.emplace_back(it->dev, it->begin);
                    m_spare.emplace_back(it->dev, it->copied);
                    m_spare.emplace_back(it->dev, it->done);
                    it = m_inflight.erase(it);
                    continue;
                }
            }
            if (m_verbose) {
                float ms = 0.f;
                vkmr_hip_event_elapsed_ms(it->dev, it->begin, it->done, &ms);
                std::cout << "Mapping for slice #" << it->sub.Number() << " (" << it->sub.Reserved() << " item(s); "
                          << it->batch.Size() << " byte(s)) finished in " << ms << "ms." << std::endl;
            }
            m_spare.emplace_back(it->dev, it->begin);
            m_spare.emplace_back(it->dev, it->copied);
            m_spare.emplace_back(it->dev, it->done);
            out.push_back(std::move(it->sub));
            it = m_inflight.erase(it);   // the batch goes back to its pool here
        }
        return out;
    }

    bool m_verbose;
    bool m_send_sizes;   // describe a batch to the device by its 16-bit sizes when they suffice (HipConfig::send_sizes)
    bool m_failed = false;
    std::vector<Mapping> m_inflight;
    std::vector<std::pair<int, vkmr_event>> m_spare;
};

}  // namespace

std::unique_ptr<Mappings> Mappings::New(bool verbose, bool send_sizes) { return std::unique_ptr<Mappings>(new MappingsImpl(verbose, send_sizes)); }

}  // namespace vkmr
