// reductions.cpp -- in-flight reductions and the final combine (reference
// src/vkmr/Reductions.cpp:621-713).  A reduction is one vkmr_hip_reduce_async on the
// op's stream followed by a 32-byte copy of the root into pinned host memory (the
// reference's vkCmdCopyBuffer, Reductions.cpp:537-540) and an event.
#include <iostream>
#include <map>

#include "cpu_sha256d.hpp"
#include "ops.hpp"

namespace vkmr {
namespace {

struct Reduction {
    Slice slice;
    void* scratch = nullptr;
    vkmr_digest* root_dev = nullptr;
    vkmr_digest* root_host = nullptr;
    vkmr_event begin = nullptr, done = nullptr;
    int dev = -1;
};

class ReductionsImpl : public Reductions {
public:
    ReductionsImpl(int combine_device, bool verbose) : m_combine_dev(combine_device), m_verbose(verbose) {}
    ~ReductionsImpl() override
    {
        for (auto& r : m_inflight) {
            vkmr_hip_event_wait(r.dev, r.done);
            Free(r);
        }
    }

    HipResult Reduce(slice_type&& slice, uint32_t height, vkmr_stream stream) override
    {
        if (!slice || slice.Count() == 0) return VKMR_OK;
        Reduction r;
        r.dev = slice.Device();
        void *scratch = nullptr, *rd = nullptr, *rh = nullptr;
        HipResult st = vkmr_hip_device_alloc(r.dev, vkmr_hip_reduce_scratch_bytes(slice.Count()), &scratch);
        if (st == VKMR_OK) st = vkmr_hip_device_alloc(r.dev, sizeof(vkmr_digest), &rd);
        if (st == VKMR_OK) st = vkmr_hip_host_alloc(sizeof(vkmr_digest), &rh);
        if (st == VKMR_OK) st = vkmr_hip_event_create(r.dev, &r.begin);
        if (st == VKMR_OK) st = vkmr_hip_event_create(r.dev, &r.done);
        r.scratch = scratch;
        r.root_dev = static_cast<vkmr_digest*>(rd);
        r.root_host = static_cast<vkmr_digest*>(rh);
        if (st == VKMR_OK) st = vkmr_hip_event_record(r.dev, r.begin, stream);
        if (st == VKMR_OK) st = vkmr_hip_reduce_async(r.dev, stream, slice.Cells(), slice.Count(), height, r.scratch, r.root_dev);
        if (st == VKMR_OK) st = vkmr_hip_memcpy_d2h_async(r.dev, stream, r.root_host, r.root_dev, sizeof(vkmr_digest));
        if (st == VKMR_OK) st = vkmr_hip_event_record(r.dev, r.done, stream);
        if (st != VKMR_OK) {
            std::cerr << "Failed to dispatch a reduction: " << vkmr_hip_last_error() << std::endl;
            Free(r);
            return st;
        }
        r.slice = std::move(slice);
        m_inflight.push_back(std::move(r));
        return VKMR_OK;
    }

    void Update() override { Retire(false); }

    ISha256D::out_type WaitFor() override
    {
        Retire(true);
        if (m_results.empty()) return "";
        if (m_results.size() == 1) return digest_words_to_hex(m_results.begin()->second.data);
        // slice roots in slice order 1..n (reference Reductions.cpp:703-712); any gap is a failure
        std::vector<vkmr_digest> roots;
        uint32_t expect = 1;
        for (const auto& kv : m_results) {
            if (kv.first != expect++) return "";
            roots.push_back(kv.second);
        }
        vkmr_digest top;
        if (vkmr_hip_combine(m_combine_dev, roots.data(), (uint32_t)roots.size(), &top) != VKMR_OK) {
            std::cerr << "Failed to combine the slice roots: " << vkmr_hip_last_error() << std::endl;
            return "";
        }
        return digest_words_to_hex(top.data);
    }

private:
    void Free(Reduction& r)
    {
        vkmr_hip_device_free(r.dev, r.scratch);
        vkmr_hip_device_free(r.dev, r.root_dev);
        vkmr_hip_host_free(r.root_host);
        vkmr_hip_event_destroy(r.dev, r.begin);
        vkmr_hip_event_destroy(r.dev, r.done);
        r.scratch = nullptr; r.root_dev = nullptr; r.root_host = nullptr; r.begin = r.done = nullptr;
    }

    void Retire(bool block)
    {
        for (auto it = m_inflight.begin(); it != m_inflight.end();) {
            HipResult st = vkmr_hip_event_query(it->dev, it->done);
            if (st == VKMR_NOT_READY && block) st = vkmr_hip_event_wait(it->dev, it->done);
            if (st == VKMR_NOT_READY) {
                ++it;
                continue;
            }
            if (m_verbose) {
                float ms = 0.f;
                vkmr_hip_event_elapsed_ms(it->dev, it->begin, it->done, &ms);
                std::cout << "Reduction #" << it->slice.Number() << " finished in " << ms << "ms." << std::endl;
                std::cout << "#" << it->slice.Number() << ":" << digest_words_to_hex(it->root_host->data) << std::endl;
            }
            m_results[it->slice.Number()] = *it->root_host;
            Free(*it);
            it = m_inflight.erase(it);   // the slice's memory is released here
        }
    }

    int m_combine_dev;
    bool m_verbose;
    std::vector<Reduction> m_inflight;
    std::map<uint32_t, vkmr_digest> m_results;
};

}  // namespace

std::unique_ptr<Reductions> Reductions::New(int combine_device, bool verbose)
{
    return std::unique_ptr<Reductions>(new ReductionsImpl(combine_device, verbose));
}

}  // namespace vkmr
