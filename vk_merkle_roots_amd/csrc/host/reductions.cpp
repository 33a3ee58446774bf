// reductions.cpp -- in-flight reductions and the final combine (reference
// src/vkmr/Reductions.cpp:621-713).  A reduction is one vkmr_hip_reduce_async on the
// op's stream, its root written into the device's root array, a 32-byte copy of it
// into the pinned host mirror (the reference's vkCmdCopyBuffer, Reductions.cpp:537-540)
// and an event.  See ops.hpp for the layout of the root arrays and the pools.
#include <algorithm>
#include <cstring>
#include <iostream>
#include <string>

#include "cpu_sha256d.hpp"
#include "ops.hpp"

namespace vkmr {
namespace {

// What one reduction in flight needs besides its slice; recycled through PerDevice::spare.
struct Resources {
    void* scratch = nullptr;
    vkmr_event begin = nullptr, done = nullptr;
};

struct Reduction {
    Slice slice;
    Resources res;
    size_t device_index = 0;   // index into m_devs
    uint32_t slot = 0;         // entry of the device's root array
};

struct PerDevice {
    int dev = -1;
    vkmr_stream stream = nullptr;        // the stream of this device's reductions (as given to Reduce)
    vkmr_digest* roots_dev = nullptr;    // HBM: root of the device's j-th slice at [j]
    vkmr_digest* roots_host = nullptr;   // pinned mirror
    uint32_t roots_cap = 0, roots_used = 0;
    std::vector<Resources> spare;
};

class ReductionsImpl : public Reductions {
public:
    ReductionsImpl(std::vector<int> devices, size_t capacity, bool verbose)
        : m_capacity(capacity), m_scratch_bytes(vkmr_hip_reduce_scratch_bytes(capacity)), m_verbose(verbose)
    {
        for (int d : devices) {
            PerDevice pd;
            pd.dev = d;
            m_devs.push_back(pd);
        }
        // What every device needs for its first reduction is taken now, before slices and batches fill HBM:
        // the root array and one scratch + event set.  Later reductions re-use that set (or wait for it), so
        // a reduction can always be dispatched, however full the device is by then.
        for (size_t di = 0; di < m_devs.size(); ++di) {
            Resources r;
            HipResult st = EnsureRoots(di, 0);
            if (st == VKMR_OK) st = NewResources(m_devs[di].dev, &r);
            if (st != VKMR_OK) {
                std::cerr << "Failed to prepare reductions on device " << m_devs[di].dev << ": " << vkmr_hip_last_error() << std::endl;
                m_failed = true;
                break;
            }
            m_devs[di].spare.push_back(r);
        }
    }

    ~ReductionsImpl() override
    {
        for (auto& r : m_inflight) {
            vkmr_hip_event_wait(m_devs[r.device_index].dev, r.res.done);
            m_devs[r.device_index].spare.push_back(r.res);
        }
        m_inflight.clear();   // slices go back to their pool
        for (auto& d : m_devs) {
            for (auto& res : d.spare) FreeResources(d.dev, res);
            vkmr_hip_device_free(d.dev, d.roots_dev);
            vkmr_hip_host_free(d.roots_host);
        }
        if (m_comm) vkmr_hip_comm_destroy(m_comm);
        for (auto& pd : m_proof_dev) vkmr_hip_device_free(pd.first, pd.second);
        vkmr_hip_host_free(m_proof_host);
    }

    HipResult Reduce(slice_type&& slice, uint32_t height, vkmr_stream stream) override
    {
        if (!slice || slice.Count() == 0) return VKMR_OK;
        const HipResult st = Dispatch(slice, height, stream);
        if (st != VKMR_OK) {
            std::cerr << "Failed to dispatch a reduction: " << vkmr_hip_last_error() << std::endl;
            m_failed = true;   // a slice without a root: the run has no root either
        }
        return st;
    }

    void Update() override { Retire(false, (size_t)-1); }

    bool WaitOne() override
    {
        if (m_inflight.empty()) return false;
        Retire(true, m_inflight.size() - 1);
        return true;
    }

    size_t InFlight() const override { return m_inflight.size(); }
    size_t Allocations() const override { return m_allocations; }
    bool Ok() const override { return !m_failed; }

    void RequestProof(uint64_t leaf_index) override
    {
        if (m_proofs.size() >= kMaxProofs) {
            std::cerr << "At most " << kMaxProofs << " Merkle proofs per run: leaf " << leaf_index << " is not proved." << std::endl;
            return;
        }
        ProofReq r;
        r.leaf = leaf_index;
        r.host_at = m_proofs.size() * kProofRecord;
        m_proofs.push_back(r);
    }

    std::vector<std::string> ProofLines() const override
    {
        std::vector<std::string> lines;
        for (const ProofReq& r : m_proofs) {
            if (!r.done) {
                lines.push_back("proof: leaf " + std::to_string(r.leaf) + " is not in the stream (or the run failed)");
                continue;
            }
            lines.push_back("proof: leaf " + std::to_string(r.leaf) + " " + digest_words_to_hex(r.leaf_digest.data));
            // bit l of the node index tells on which side the path node sits at level l; inside the slice the node index is the
            // leaf's offset, above it the slice's position
            for (size_t l = 0; l < r.path.size(); ++l) {
                const uint64_t idx = l < r.slice_levels ? r.offset : (uint64_t)(r.slice_number - 1);
                const bool path_is_right = (idx >> (l < r.slice_levels ? l : l - r.slice_levels)) & 1ull;
                lines.push_back("proof: level " + std::to_string(l) + (path_is_right ? " sibling-on-left " : " sibling-on-right ") +
                                digest_words_to_hex(r.path[l].data));
            }
        }
        return lines;
    }

    ISha256D::out_type WaitFor() override
    {
        Retire(true, 0);
        if (m_failed || m_dispatched == 0) return "";
        // every slice 1..n must have been reduced: n roots for slice numbers up to n
        // (reference: roots looked up by slice number 1..n, Reductions.cpp:703-712)
        if (m_retired != m_dispatched || m_last_number != m_dispatched) {
            std::cerr << "Reduced " << m_retired << " of " << m_last_number << " slice(s); no root." << std::endl;
            return "";
        }
        if (m_dispatched == 1) {   // Reductions.cpp:692-701
            FinishProof(nullptr, 0);
            return digest_words_to_hex(m_devs[0].roots_host[0].data);
        }
        vkmr_digest top;
        if (Combine(&top) != VKMR_OK) {
            std::cerr << "Failed to combine the slice roots: " << vkmr_hip_last_error() << std::endl;
            return "";
        }
        return digest_words_to_hex(top.data);
    }

private:
    void FreeResources(int dev, Resources& r)
    {
        vkmr_hip_device_free(dev, r.scratch);
        vkmr_hip_event_destroy(dev, r.begin);
        vkmr_hip_event_destroy(dev, r.done);
        r = Resources();
    }

    // Scratch + events for one more reduction on device `di`: recycled, freshly allocated, or --
    // when HBM has no room -- taken over from the oldest reduction once it has retired.
    HipResult Acquire(size_t di, Resources* out)
    {
        PerDevice& d = m_devs[di];
        for (;;) {
            if (!d.spare.empty()) {
                *out = d.spare.back();
                d.spare.pop_back();
                return VKMR_OK;
            }
            // a reduction of this device in flight will hand its set back: prefer that to another allocation
            // once a few sets exist (they are per-reduction, and reductions of one device share a stream anyway)
            const size_t mine = (size_t)std::count_if(m_inflight.begin(), m_inflight.end(), [&](const Reduction& r) { return r.device_index == di; });
            if (mine >= 4) {
                WaitOne();
                continue;
            }
            Resources r;
            const HipResult st = NewResources(d.dev, &r);
            if (st == VKMR_OK) {
                *out = r;
                return VKMR_OK;
            }
            if (st != VKMR_ERR_OOM || !WaitOne()) return st;   // with a set reserved per device, there is always one to wait for
        }
    }

    HipResult NewResources(int dev, Resources* out)
    {
        Resources r;
        HipResult st = vkmr_hip_device_alloc(dev, m_scratch_bytes, &r.scratch);
        if (st == VKMR_OK) st = vkmr_hip_event_create(dev, &r.begin);
        if (st == VKMR_OK) st = vkmr_hip_event_create(dev, &r.done);
        if (st != VKMR_OK) {
            FreeResources(dev, r);
            return st;
        }
        ++m_allocations;
        *out = r;
        return VKMR_OK;
    }

    // Room for entry `slot` of the device's root array (grows by doubling; the reductions of this
    // device are drained first, their roots copied over -- rare: 4096 slices fit before the first time).
    HipResult EnsureRoots(size_t di, uint32_t slot)
    {
        PerDevice& d = m_devs[di];
        if (slot < d.roots_cap) return VKMR_OK;
        uint32_t cap = d.roots_cap ? d.roots_cap : 4096u;
        while (cap <= slot) cap *= 2u;
        void *nd = nullptr, *nh = nullptr;
        HipResult st = vkmr_hip_device_alloc(d.dev, (size_t)cap * sizeof(vkmr_digest), &nd);
        if (st == VKMR_OK) st = vkmr_hip_host_alloc((size_t)cap * sizeof(vkmr_digest), &nh);
        if (st != VKMR_OK) {
            vkmr_hip_device_free(d.dev, nd);
            return st;
        }
        if (d.roots_cap) {
            while (std::any_of(m_inflight.begin(), m_inflight.end(), [&](const Reduction& r) { return r.device_index == di; })) WaitOne();
            std::memcpy(nh, d.roots_host, (size_t)d.roots_cap * sizeof(vkmr_digest));
            st = vkmr_hip_memcpy_h2d_async(d.dev, d.stream, nd, nh, (size_t)d.roots_cap * sizeof(vkmr_digest));
            if (st == VKMR_OK) st = vkmr_hip_stream_sync(d.dev, d.stream);
            vkmr_hip_device_free(d.dev, d.roots_dev);
            vkmr_hip_host_free(d.roots_host);
        }
        d.roots_dev = static_cast<vkmr_digest*>(nd);
        d.roots_host = static_cast<vkmr_digest*>(nh);
        d.roots_cap = cap;
        return st;
    }

    // The reduction of a slice that holds leaves whose proofs were asked for: the same launches, with the kernels that write
    // the siblings of those leaves' path nodes as they hash them (vkmr_hip_reduce_proofs_async; the reference's to-do,
    // README.md:118-120: "write out the intermediate values during reduction").  The leaf digests and the sibling rows
    // are copied to the host on the reduction's stream.
    HipResult ReduceAndProve(PerDevice& d, const slice_type& slice, uint32_t height, void* scratch, vkmr_stream stream, vkmr_digest* root_dev)
    {
        std::vector<ProofReq*> here;
        std::vector<uint64_t> offsets;
        for (ProofReq& r : m_proofs) {
            if (r.seen || slice.Number() != r.leaf / m_capacity + 1) continue;
            r.seen = true;
            r.offset = r.leaf % m_capacity;
            if (r.offset >= slice.Count()) continue;   // past the end of the stream: reported by ProofLines()
            r.slice_levels = height;
            r.slice_number = slice.Number();
            here.push_back(&r);
            offsets.push_back(r.offset);
        }
        if (here.empty()) return vkmr_hip_reduce_async(d.dev, stream, slice.Cells(), slice.Count(), height, scratch, root_dev);
        void* sib = nullptr;
        HipResult st = vkmr_hip_device_alloc(d.dev, (here.size() * (size_t)height + 1) * sizeof(vkmr_digest), &sib);
        if (st == VKMR_OK) m_proof_dev.push_back(std::make_pair(d.dev, sib));
        if (st == VKMR_OK && !m_proof_host) {
            void* h = nullptr;
            st = vkmr_hip_host_alloc(kMaxProofs * kProofRecord * sizeof(vkmr_digest), &h);   // per proof one record: the leaf + its siblings inside the slice (those above arrive in Combine's own buffer)
            m_proof_host = static_cast<vkmr_digest*>(h);
        }
        if (st == VKMR_OK)
            st = vkmr_hip_reduce_proofs_async(d.dev, stream, slice.Cells(), slice.Count(), height, scratch, root_dev, offsets.data(), (uint32_t)offsets.size(),
                                              static_cast<vkmr_digest*>(sib));
        for (size_t q = 0; q < here.size() && st == VKMR_OK; ++q) {
            ProofReq& r = *here[q];
            st = vkmr_hip_memcpy_d2h_async(d.dev, stream, m_proof_host + r.host_at, slice.Cells() + r.offset, sizeof(vkmr_digest));
            if (st == VKMR_OK && height)
                st = vkmr_hip_memcpy_d2h_async(d.dev, stream, m_proof_host + r.host_at + 1, static_cast<vkmr_digest*>(sib) + q * (size_t)height,
                                               (size_t)height * sizeof(vkmr_digest));
            r.dispatched = (st == VKMR_OK);
        }
        return st;
    }

    HipResult Dispatch(slice_type& slice, uint32_t height, vkmr_stream stream)
    {
        size_t di = 0;
        while (di < m_devs.size() && m_devs[di].dev != slice.Device()) ++di;
        if (di == m_devs.size() || slice.Number() == 0) return VKMR_ERR_INVALID;
        // The combine reads the roots as dealt: slice k on device (k - 1) % D, entry (k - 1) / D of that device's array
        // (WaitFor, roots_in_slice_order).  A slice that sits anywhere else -- an allocator that fell back to another
        // device, a different device order -- would put two roots into one cell or the cells into the wrong order, and
        // only the COUNT of roots is checked later: refuse it here and mark the run failed.
        if (di != (size_t)((slice.Number() - 1) % m_devs.size())) {
            std::cerr << "Slice #" << slice.Number() << " lives on device index " << di << ", the deal puts it on "
                      << (slice.Number() - 1) % m_devs.size() << ": its root would land in the wrong place." << std::endl;
            m_failed = true;
            return VKMR_ERR_INVALID;
        }
        PerDevice& d = m_devs[di];
        d.stream = stream;
        Reduction r;
        r.device_index = di;
        r.slot = (uint32_t)((slice.Number() - 1) / m_devs.size());
        HipResult st = EnsureRoots(di, r.slot);
        if (st == VKMR_OK) st = Acquire(di, &r.res);
        if (st != VKMR_OK) return st;
        st = vkmr_hip_event_record(d.dev, r.res.begin, stream);
        if (st == VKMR_OK) st = ReduceAndProve(d, slice, height, r.res.scratch, stream, d.roots_dev + r.slot);
        if (st == VKMR_OK) st = vkmr_hip_memcpy_d2h_async(d.dev, stream, d.roots_host + r.slot, d.roots_dev + r.slot, sizeof(vkmr_digest));
        if (st == VKMR_OK) st = vkmr_hip_event_record(d.dev, r.res.done, stream);
        if (st != VKMR_OK) {
            d.spare.push_back(r.res);
            return st;
        }
        if (r.slot + 1 > d.roots_used) d.roots_used = r.slot + 1;
        if (slice.Number() > m_last_number) m_last_number = slice.Number();
        ++m_dispatched;
        r.slice = std::move(slice);
        m_inflight.push_back(std::move(r));
        return VKMR_OK;
    }

    // Retires finished reductions, oldest first; with `block` waits until at most `keep` remain.
    void Retire(bool block, size_t keep)
    {
        for (auto it = m_inflight.begin(); it != m_inflight.end();) {
            PerDevice& d = m_devs[it->device_index];
            HipResult st = vkmr_hip_event_query(d.dev, it->res.done);
            if (st == VKMR_NOT_READY && block && m_inflight.size() > keep) st = vkmr_hip_event_wait(d.dev, it->res.done);
            if (st == VKMR_NOT_READY) {
                ++it;
                continue;
            }
            if (st < 0) {   // the device reported an error: this slice has no root
                std::cerr << "Reduction #" << it->slice.Number() << " failed: " << vkmr_hip_last_error() << std::endl;
                m_failed = true;
            } else {
                ++m_retired;
                if (m_verbose) {
                    float ms = 0.f;
                    vkmr_hip_event_elapsed_ms(d.dev, it->res.begin, it->res.done, &ms);
                    std::cout << "Reduction #" << it->slice.Number() << " finished in " << ms << "ms." << std::endl;
                    std::cout << "#" << it->slice.Number() << ":" << digest_words_to_hex(d.roots_host[it->slot].data) << std::endl;
                }
            }
            d.spare.push_back(it->res);
            it = m_inflight.erase(it);   // the slice's memory returns to its pool here
        }
    }

    // Everything of the proofs has reached the host (the streams were synchronised): assemble the paths.  upper: per
    // dispatched proof (in request order) `upper_levels` siblings above the slices.
    void FinishProof(const vkmr_digest* upper, uint32_t upper_levels)
    {
        if (!m_proof_host || m_failed) return;
        size_t nth = 0;
        for (ProofReq& r : m_proofs) {
            if (!r.dispatched) continue;
            r.leaf_digest = m_proof_host[r.host_at];
            r.path.assign(m_proof_host + r.host_at + 1, m_proof_host + r.host_at + 1 + r.slice_levels);
            for (uint32_t l = 0; l < upper_levels; ++l) r.path.push_back(upper[nth * upper_levels + l]);
            r.done = true;
            ++nth;
        }
    }

    // Root over all slice roots, in slice order, on the first device.
    HipResult Combine(vkmr_digest* top)
    {
        const uint32_t total = m_dispatched;
        const size_t ndev = m_devs.size();
        PerDevice& d0 = m_devs[0];
        void *ordered = nullptr, *scratch = nullptr, *final_dev = nullptr, *final_host = nullptr;
        std::vector<void*> gathered(ndev, nullptr);
        HipResult st = VKMR_OK;
        const vkmr_digest* roots = d0.roots_dev;
        if (ndev > 1) {
            // ONE all-gather of the per-device root arrays over RCCL, then slice order on device 0
            uint32_t per_rank = 0;
            for (const auto& d : m_devs) per_rank = std::max(per_rank, d.roots_used);
            std::vector<vkmr_stream> streams;
            std::vector<const vkmr_digest*> mine;
            std::vector<vkmr_digest*> all;
            if (!m_comm) {
                std::vector<int> ids;
                for (const auto& d : m_devs) ids.push_back(d.dev);
                st = vkmr_hip_comm_init_all(ids.data(), (int)ids.size(), &m_comm);
            }
            for (size_t i = 0; i < ndev && st == VKMR_OK; ++i) {
                PerDevice& d = m_devs[i];
                if (!d.stream) {   // a device that got no slice still takes part in the collective
                    st = vkmr_hip_stream_create(d.dev, &d.stream);
                    m_own_streams.push_back(i);
                }
                if (st == VKMR_OK) st = EnsureRoots(i, per_rank - 1);   // every rank sends per_rank cells
                if (st == VKMR_OK) st = vkmr_hip_device_alloc(d.dev, ndev * (size_t)per_rank * sizeof(vkmr_digest), &gathered[i]);
                streams.push_back(d.stream);
                mine.push_back(d.roots_dev);
                all.push_back(static_cast<vkmr_digest*>(gathered[i]));
            }
            if (st == VKMR_OK) st = vkmr_hip_gather_roots_async(m_comm, streams.data(), mine.data(), per_rank, all.data());
            if (st == VKMR_OK) st = vkmr_hip_device_alloc(d0.dev, (size_t)total * sizeof(vkmr_digest), &ordered);
            if (st == VKMR_OK)
                st = vkmr_hip_roots_in_slice_order_async(d0.dev, d0.stream, all[0], (uint32_t)ndev, per_rank, total, static_cast<vkmr_digest*>(ordered));
            roots = static_cast<const vkmr_digest*>(ordered);
        }
        if (st == VKMR_OK) st = vkmr_hip_device_alloc(d0.dev, vkmr_hip_reduce_scratch_bytes(total), &scratch);
        if (st == VKMR_OK) st = vkmr_hip_device_alloc(d0.dev, sizeof(vkmr_digest), &final_dev);
        if (st == VKMR_OK) st = vkmr_hip_host_alloc(sizeof(vkmr_digest), &final_host);
        // the proofs' upper parts: the paths of the leaves' slices among the slice roots -- written by the combine itself
        // (same tree, same launches)
        void* up_sib = nullptr;
        uint32_t up_levels = 0;
        std::vector<uint64_t> positions;
        for (const ProofReq& r : m_proofs)
            if (r.dispatched) positions.push_back((uint64_t)r.slice_number - 1);
        vkmr_digest* upper_host = nullptr;
        if (st == VKMR_OK && !positions.empty()) {
            up_levels = 1;
            while (((uint64_t)total + ((1ull << up_levels) - 1ull)) >> up_levels > 1) ++up_levels;
            st = vkmr_hip_device_alloc(d0.dev, positions.size() * (size_t)up_levels * sizeof(vkmr_digest), &up_sib);
            if (st == VKMR_OK)
                st = vkmr_hip_reduce_proofs_async(d0.dev, d0.stream, roots, total, up_levels, scratch, static_cast<vkmr_digest*>(final_dev), positions.data(),
                                                  (uint32_t)positions.size(), static_cast<vkmr_digest*>(up_sib));
            if (st == VKMR_OK) {
                void* h = nullptr;
                st = vkmr_hip_host_alloc(positions.size() * (size_t)up_levels * sizeof(vkmr_digest), &h);
                upper_host = static_cast<vkmr_digest*>(h);
            }
            if (st == VKMR_OK)
                st = vkmr_hip_memcpy_d2h_async(d0.dev, d0.stream, upper_host, up_sib, positions.size() * (size_t)up_levels * sizeof(vkmr_digest));
        } else if (st == VKMR_OK) {
            st = vkmr_hip_combine_async(d0.dev, d0.stream, roots, total, scratch, static_cast<vkmr_digest*>(final_dev));
        }
        if (st == VKMR_OK) st = vkmr_hip_memcpy_d2h_async(d0.dev, d0.stream, final_host, final_dev, sizeof(vkmr_digest));
        if (st == VKMR_OK) st = vkmr_hip_stream_sync(d0.dev, d0.stream);
        for (size_t i = 1; i < ndev && st == VKMR_OK; ++i)   // the other ranks' side of the gather
            if (m_devs[i].stream && gathered[i]) st = vkmr_hip_stream_sync(m_devs[i].dev, m_devs[i].stream);
        if (st == VKMR_OK) std::memcpy(top, final_host, sizeof(vkmr_digest));
        if (st == VKMR_OK) FinishProof(upper_host, up_levels);
        vkmr_hip_host_free(upper_host);
        vkmr_hip_device_free(d0.dev, up_sib);
        for (size_t i = 0; i < ndev; ++i) vkmr_hip_device_free(m_devs[i].dev, gathered[i]);
        vkmr_hip_device_free(d0.dev, ordered);
        vkmr_hip_device_free(d0.dev, scratch);
        vkmr_hip_device_free(d0.dev, final_dev);
        vkmr_hip_host_free(final_host);
        for (size_t i : m_own_streams) {
            vkmr_hip_stream_destroy(m_devs[i].dev, m_devs[i].stream);
            m_devs[i].stream = nullptr;
        }
        m_own_streams.clear();
        return st;
    }

    size_t m_capacity, m_scratch_bytes, m_allocations = 0;
    bool m_verbose;
    bool m_failed = false;
    uint32_t m_dispatched = 0, m_retired = 0, m_last_number = 0;
    std::vector<PerDevice> m_devs;
    std::vector<Reduction> m_inflight;
    std::vector<size_t> m_own_streams;
    vkmr_comm m_comm = nullptr;
    // the requested Merkle proofs
    static constexpr size_t kMaxProofs = 16;      // what one reduction writes in its pass (vkmr_hip_reduce_proofs_async)
    static constexpr size_t kProofRecord = 66;    // pinned digests per proof: the leaf + up to 64 levels inside its slice (+ 1 spare)
    struct ProofReq {
        uint64_t leaf = 0, offset = 0;            // stream index; index inside its slice
        uint32_t slice_number = 0, slice_levels = 0;
        bool seen = false, dispatched = false, done = false;
        size_t host_at = 0;                       // this proof's record in m_proof_host
        vkmr_digest leaf_digest{};
        std::vector<vkmr_digest> path;
    };
    std::vector<ProofReq> m_proofs;
    vkmr_digest* m_proof_host = nullptr;          // pinned: per proof the leaf digest and the siblings inside its slice
    std::vector<std::pair<int, void*>> m_proof_dev;   // device sibling arrays (device, pointer), freed with the object
};

}  // namespace

std::unique_ptr<Reductions> Reductions::New(std::vector<int> devices, size_t capacity, bool verbose)
{
    return std::unique_ptr<Reductions>(new ReductionsImpl(std::move(devices), capacity, verbose));
}

}  // namespace vkmr
