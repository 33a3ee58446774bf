// hip_sha256d.cpp -- see hip_sha256d.hpp.  Control flow follows the reference's stream
// processor (src/vkmr/SHA-256vk.cpp:288-429) with two simplifications HIP allows: a
// sub-slice is a pointer offset, so strings go straight into the batch (no buffered
// vector and no aligned reservation size, SHA-256vk.cpp:338-342), and a full pipeline
// blocks on its oldest mapping instead of failing.
#include "hip_sha256d.hpp"

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>

#include "fork_join.hpp"
#include "timing.hpp"
#include "util.hpp"

namespace vkmr {

HipConfig HipConfig::FromEnv()
{
    HipConfig c;
    if (const char* e = getenv("VKMR_SLICE_LOG2")) {
        c.slice_log2 = (uint32_t)atoi(e);
        c.slice_log2_given = true;
        if (c.slice_log2 < 1) c.slice_log2 = 1;
        if (c.slice_log2 > 40) c.slice_log2 = 40;   // clamped again to what the devices hold (ChooseSliceLog2)
    }
    if (const char* e = getenv("VKMR_BATCH_MB")) c.batch_bytes = (size_t)atol(e) << 20;
    if (const char* e = getenv("VKMR_BATCH_BYTES")) c.batch_bytes = (size_t)atol(e);
    if (const char* e = getenv("VKMR_BATCH_MAX_MB")) c.batch_bytes_max = (size_t)atol(e) << 20;
    if (const char* e = getenv("VKMR_MAX_INFLIGHT")) c.max_inflight = (size_t)atol(e);
    if (const char* e = getenv("VKMR_SLICE_BUDGET")) c.slice_budget = (size_t)atol(e);
    if (const char* e = getenv("VKMR_VERBOSE")) c.verbose = atoi(e) != 0;
    if (const char* e = getenv("VKMR_PROOF_INDEX")) {   // "a" or "a,b,c": leaves to prove
        for (const char* p = e; *p;) {
            char* end = nullptr;
            const long long v = strtoll(p, &end, 10);
            if (end == p) break;
            if (v >= 0) c.proof_indices.push_back((unsigned long long)v);
            p = (*end == ',') ? end + 1 : end;
            if (*end && *end != ',') break;
        }
    }
    if (const char* e = getenv("VKMR_SEND_METADATA")) c.send_sizes = atoi(e) == 0;
#ifdef VKMR_EXPERIMENTS
    if (const char* e = getenv("VKMR_DEVICE_SPLIT")) c.device_split = atoi(e) != 0;   // experiments build only (include/vkmr_hip_experiments.h)
#endif
    if (const char* e = getenv("VKMR_PACK_STREAM")) c.pack_stream = atoi(e) < 0 ? -1 : (atoi(e) != 0);
    if (const char* e = getenv("VKMR_PACK_THREADS")) c.pack_threads = (unsigned)atoi(e);
    if (c.pack_threads == 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        c.pack_threads = hw == 0 ? 1u : (hw > 16u ? 16u : hw);
    }
    if (c.batch_bytes < 4096) c.batch_bytes = 4096;
    // a packed batch addresses its data by 32-bit word index (vkmr_metadata::start)
    if (c.batch_bytes > (size_t)0xFFFFFFFFull * 4u) c.batch_bytes = (size_t)0xFFFFFFFFull * 4u;
    if (c.batch_bytes_max > (size_t)0xFFFFFFFFull * 4u) c.batch_bytes_max = (size_t)0xFFFFFFFFull * 4u;
    if (c.batch_bytes_max < c.batch_bytes) c.batch_bytes_max = c.batch_bytes;
    if (c.max_inflight < 1) c.max_inflight = 1;
    return c;
}

int HipSha256D::Count() const
{
    if (m_count < 0) {
        int n = 0;
        m_count = (vkmr_hip_device_count(&n) == VKMR_OK) ? n : 0;
    }
    return m_count;
}

std::vector<ISha256D::name_type> HipSha256D::Available() const
{
    std::vector<ISha256D::name_type> names;
    for (int i = 0; i < Count(); ++i) names.push_back("hip:" + std::to_string(i));
    if (Count() > 1) names.push_back("hip:all");
    return names;
}

// The reference selects a GPU by its Vulkan deviceName (src/vkmr/SHA-256vk.cpp:219-229; .vscode/launch.json:12
// hard-codes one).  Here devices are listed as "hip:<n>" (eight MI355X of a node share one marketing name), and a
// marketing name is accepted as an alias of the first device that carries it.
int HipSha256D::IndexOf(const ISha256D::name_type& name) const
{
    if (name.size() > 4 && name.compare(0, 4, "hip:") == 0 && name.find_first_not_of("0123456789", 4) == std::string::npos) {
        const long i = atol(name.c_str() + 4);
        return i < Count() ? (int)i : -1;
    }
    for (int i = 0; i < Count(); ++i) {
        char devname[256] = "";
        if (vkmr_hip_device_name(i, devname, sizeof devname) == VKMR_OK && name == devname) return i;
    }
    return -1;
}

std::string HipSha256D::Describe(const ISha256D::name_type& name) const
{
    if (name.compare(0, 4, "hip:") != 0 || name == "hip:all") return name == "hip:all" ? " (slices dealt over every GPU)" : "";
    char devname[256] = "";
    const int i = IndexOf(name);
    if (i < 0 || vkmr_hip_device_name(i, devname, sizeof devname) != VKMR_OK || !devname[0]) return "";
    return std::string(" (\"") + devname + "\")";
}

bool HipSha256D::Has(const ISha256D::name_type& name) const
{
    if (name == "CPU") return false;   // the other backend's name: no reason to wake the GPU runtime
    if (name == "hip:all") return Count() > 1;
    return IndexOf(name) >= 0;
}

std::unique_ptr<HipSha256D::Instance> HipSha256D::Get(const ISha256D::name_type& name, const HipConfig& cfg)
{
    std::vector<int> devs;
    if (name == "hip:all") {
        for (int i = 0; i < Count(); ++i) devs.push_back(i);
    } else {
        devs.push_back(IndexOf(name));
    }
    return std::unique_ptr<Instance>(new Instance(name, devs, cfg));
}

HipSha256D::Instance::Instance(const std::string& name, std::vector<int> devices, const HipConfig& cfg)
    : ISha256D(name), m_cfg(cfg), m_device_ids(devices), m_ok(true)
{
    for (int d : devices) {
        PerDevice pd;
        pd.dev = d;
        // an input of known size that one batch holds gets a batch of that size (pinning 80 MiB for a few strings is 13 ms)
        size_t batch_bytes = cfg.batch_bytes;
        if (cfg.expected_input_bytes > 0) {
            const uint64_t need = ((cfg.expected_input_bytes + cfg.expected_input_bytes / 4 + (1u << 16)) + ((1u << 20) - 1)) & ~(uint64_t)((1u << 20) - 1);
            if (need < batch_bytes) batch_bytes = (size_t)need;
        }
        pd.batches.reset(new Batches(d, batch_bytes, cfg.device_split, cfg.pack_stream));
        if (cfg.verbose) {
            char devname[256] = "";
            size_t free_b = 0, total_b = 0;
            int cus = 0, wave = 0;
            vkmr_hip_device_name(d, devname, sizeof devname);
            vkmr_hip_device_mem_info(d, &free_b, &total_b);
            vkmr_hip_device_geometry(d, &cus, &wave);
            std::cout << "hip:" << d << " \"" << devname << "\": " << cus << " CUs, wavefront " << wave << ", "
                      << (free_b >> 20) << " of " << (total_b >> 20) << " MiB free" << std::endl;
        }
        m_devs.push_back(std::move(pd));
    }
    // Streams, and the start-up work the runtime would otherwise do inside the first copy and the first launch (the
    // kernels loaded onto the device: 15-20 ms; the copy engine brought up: 8-20 ms; a stream: 5-15 ms).  All of it is
    // done before the constructor returns: like the reference, whose devices build their shader modules and pipelines
    // before run() starts its stopwatch (src/vkmr/Devices.cpp:225-280).  Measured on one device
    // (profiles/r03_frontend_setup_orders.txt): these calls and the pinning of the batches serialise inside the driver
    // whatever threads issue them, so one device is set up on this thread; several devices get a thread each.
    // Reductions share the map stream: a third stream is another 10-15 ms, and the GPU of a pipeline fed over PCIe is idle
    // nine tenths of the time (the two-stream overlap pays only for resident data: bench.py, two_stream_overlap).
    auto failed = [this](const char* what) {
        std::lock_guard<std::mutex> lock(m_setup_mu);
        if (m_setup_ok) m_setup_error = std::string(what) + ": " + vkmr_hip_last_error();   // the ABI's error text is per thread: carry it over
        m_setup_ok = false;
    };
    const size_t warm_bytes = (size_t)1 << 20;   // a copy the engine treats like a batch's: after a 256-byte one the first batch still took 8-19 ms to submit,
                                                 // after 64 KiB and more 0.01 ms (profiles/r03_frontend_warm_copy.txt)
    auto set_up = [failed, warm_bytes](PerDevice* p) {
        if (vkmr_hip_stream_create(p->dev, &p->map_stream) != VKMR_OK || vkmr_hip_warm_up(p->dev, p->map_stream, VKMR_WARM_KERNELS, 0) != VKMR_OK)
            return failed("map stream");
        if (vkmr_hip_stream_create(p->dev, &p->copy_stream) != VKMR_OK || vkmr_hip_warm_up(p->dev, p->copy_stream, VKMR_WARM_COPY | VKMR_WARM_KERNELS, warm_bytes) != VKMR_OK)
            return failed("copy stream");
        p->reduce_stream = p->map_stream;
    };
    for (int d : m_device_ids) {   // what the devices have free now, before this instance takes any of it (ChooseSliceLog2)
        size_t f = 0, t = 0;
        if (vkmr_hip_device_mem_info(d, &f, &t) == VKMR_OK && f < m_free_at_start) m_free_at_start = f;
    }
    StartPrefetch();   // the first device's batches, on its pool's helper thread: ready, or nearly, when the first span arrives
    if (m_devs.size() == 1) {
        set_up(&m_devs.front());
    } else {
        std::vector<std::thread> threads;
        for (auto& pd : m_devs) threads.emplace_back(set_up, &pd);
        for (auto& t : threads) t.join();
    }
    if (!m_setup_ok) {
        std::cerr << "Failed to initialise HIP streams: " << m_setup_error << std::endl;
        m_ok = false;
    }
    m_pool.reset(new ForkJoin(cfg.pack_threads > 1 ? cfg.pack_threads - 1 : 0));
    m_mappings = Mappings::New(cfg.verbose, cfg.send_sizes);
    // slices and reductions are made when the first strings arrive: their size is chosen then (EnsureGeometry)
}

// Digests per slice, as a power of two.  The counterpart of the reference's Slices<T>::SliceSize
// (src/vkmr/Slices.h:421-454: the largest power of two within the smallest of four device limits and the preferred
// 256 MiB).  On HIP the only device limit is memory:
//   fits      what the devices' free memory holds of `budget` resident slices (32 B per digest each) plus the
//             reduction scratch sets (about 12 B per digest of capacity) beside the batch pipeline's landing zones,
//             with 15 % left alone;
//   wanted    VKMR_SLICE_LOG2 when given; else, with several devices and an input of known size (stdin is a regular
//             file), the leaves that input will hold -- estimated from the average line of the first span -- dealt
//             one slice per device (the shape BASELINE's north star names); else the reference's 2^23.
// The result is min(wanted, fits); a given value that does not fit is clamped with a message instead of failing in hipMalloc.
uint32_t HipSha256D::Instance::ChooseSliceLog2(const char* first_span, size_t len, std::string* why) const
{
    const size_t free_min = m_free_at_start;   // read before the first batch was pinned: `landing` below counts every batch
    const size_t budget = m_cfg.slice_budget ? m_cfg.slice_budget : m_cfg.max_inflight + 1;
    const double landing = (double)(m_cfg.max_inflight + 2) * ((double)m_cfg.batch_bytes * 1.25);   // data + metadata zones of the pipeline
    uint32_t fits = 40;   // no memory figure from any device: nothing to clamp against (an allocation that fails is still reported)
    if (free_min != ~(size_t)0) {
        const double room = 0.85 * (double)free_min - landing;
        // slices resident at once for a candidate size: the configured budget, or -- everything staged before the first map
        // (StagePacked) -- every slice the expected leaves fill, plus the one being filled
        auto resident = [&](uint32_t log2) -> double {
            if (m_cfg.expected_leaves == 0) return (double)budget;
            return (double)((m_cfg.expected_leaves + (((uint64_t)1 << log2) - 1)) >> log2) + 1.0;
        };
        fits = 1;
        while (fits < 40 && (double)((size_t)1 << (fits + 1)) * (32.0 * resident(fits + 1) + 12.0) <= room) ++fits;
    }
    uint32_t wanted = 23;
    *why = "the reference's 256 MiB slice (src/vkmr/SHA-256vk.cpp:23)";
    if (m_cfg.slice_log2_given) {
        wanted = m_cfg.slice_log2;
        *why = "VKMR_SLICE_LOG2";
    } else if (m_device_ids.size() > 1 && m_cfg.expected_input_bytes > 0 && first_span && len > 0) {
        size_t lines = 0;
        const size_t look = len < ((size_t)4 << 20) ? len : ((size_t)4 << 20);
        for (const char* p = first_span; (p = static_cast<const char*>(memchr(p, '\n', (size_t)(first_span + look - p)))) != nullptr; ++p) ++lines;
        if (lines > 0) {
            const double avg = (double)look / (double)lines;                        // bytes per line, newline included
            const double leaves = (double)m_cfg.expected_input_bytes / avg;
            // the estimate is good to a few percent: a power-of-two stream must not tip into slices twice as large because the
            // sample's lines were a little short (a few leaves too many simply open one more, nearly empty, slice)
            const double per_dev = leaves / (double)m_device_ids.size() / 1.03;
            wanted = 1;
            while (wanted < 40 && (double)((size_t)1 << wanted) < per_dev) ++wanted;
            if (wanted < 10) wanted = 10;
            *why = "one slice per device for about " + std::to_string((unsigned long long)leaves) + " leaves on " + std::to_string(m_device_ids.size()) + " devices";
        }
    }
    if (wanted > fits) {
        if (m_cfg.slice_log2_given)
            std::cerr << "VKMR_SLICE_LOG2=" << wanted << " does not fit the device memory (" << (free_min >> 20) << " MiB free, " << budget
                      << " slices resident): using 2^" << fits << " digests per slice." << std::endl;
        *why += ", clamped to what device memory holds";
        wanted = fits;
    }
    return wanted;
}

bool HipSha256D::Instance::EnsureGeometry(const char* first_span, size_t len)
{
    if (m_geometry) return m_ok;
    m_geometry = true;
    std::string why;
    m_cfg.slice_log2 = ChooseSliceLog2(first_span, len, &why);
    if (m_cfg.verbose) std::cout << "Slices of 2^" << m_cfg.slice_log2 << " digests (" << why << ")." << std::endl;
    const size_t capacity = (size_t)1 << m_cfg.slice_log2;
    if (m_cfg.expected_leaves) {   // the budget follows the size actually chosen, not the one the caller guessed
        const size_t need = (size_t)((m_cfg.expected_leaves + capacity - 1) >> m_cfg.slice_log2) + 1;
        if (need > m_cfg.slice_budget) m_cfg.slice_budget = need;
    }
    m_slices = Slices(m_device_ids, capacity, m_cfg.slice_budget ? m_cfg.slice_budget : m_cfg.max_inflight + 1);
    m_reductions = Reductions::New(m_device_ids, capacity, m_cfg.verbose);
    if (!m_reductions->Ok()) m_ok = false;   // reported by Reductions::New
    for (unsigned long long leaf : m_cfg.proof_indices) m_reductions->RequestProof((uint64_t)leaf);
    return m_ok;
}

// The batches of the first device's pipeline are pinned on its helper thread from now on, not when the first strings
// arrive: as many as the input will fill when its size is known, else one (the rest follow when that one goes out full).
void HipSha256D::Instance::StartPrefetch()
{
    if (!m_ok || m_devs.empty()) return;
    PerDevice& pd = m_devs.front();
    size_t want = 1;
    if (m_cfg.expected_input_bytes > 0) {
        want = (size_t)((m_cfg.expected_input_bytes + m_cfg.expected_input_bytes / 16) / pd.batches->DataBytes()) + 1;
        if (want > m_cfg.max_inflight + 1) want = m_cfg.max_inflight + 1;
    }
    pd.batches->Prefetch(want);
    pd.prefetched = want > 1;
}

HipSha256D::Instance::~Instance()
{
    // staged batches and their slice views first: each holds a raw pointer to its pool (Batch::Release -> Batches::Recycle),
    // and the pools go below (a failed StagePacked / RootOfStaged leaves them here)
    m_staged.clear();
    // ops next (they hold batches and slices), then the pools and streams
    m_pool.reset();
    m_mappings.reset();
    m_reductions.reset();
    m_batch = Batch();
    m_slices = Slices();
    for (auto& pd : m_devs) {
        pd.batches.reset();
        vkmr_hip_stream_destroy(pd.dev, pd.map_stream);
        vkmr_hip_stream_destroy(pd.dev, pd.copy_stream);
        if (pd.reduce_stream != pd.map_stream) vkmr_hip_stream_destroy(pd.dev, pd.reduce_stream);
    }
}

HipSha256D::Instance::PerDevice& HipSha256D::Instance::Dev(int dev)
{
    for (auto& pd : m_devs)
        if (pd.dev == dev) return pd;
    return m_devs.front();
}

// A retired mapping adds its strings to its slice's fill count; a filled slice goes
// to reduction at once (reference SHA-256vk.cpp:321-335).
void HipSha256D::Instance::Account(std::vector<Slice>&& retired)
{
    for (auto& sub : retired) {
        Slice& slice = m_slices[sub.Number()];
        if (!slice) continue;
        slice += sub;
        if (slice.IsFilled()) {
            if (m_cfg.verbose) std::cout << "Slice #" << slice.Number() << " has been filled." << std::endl;
            const int dev = slice.Device();
            if (m_reductions->Reduce(m_slices.Remove(sub.Number()), m_cfg.slice_log2, Dev(dev).reduce_stream) != VKMR_OK) m_ok = false;
        }
    }
    if (m_mappings->Failed()) m_ok = false;
}

// Blocks until the oldest piece of in-flight work has retired and given its buffers back
// (the reference's first to-do: "block ... and re-use the associated batch (or slice),
// rather than halting", README.md:113).  false: nothing is in flight, waiting cannot help.
bool HipSha256D::Instance::WaitForMemory()
{
    if (m_reductions->WaitOne()) return true;
    if (m_mappings->InFlight()) {
        Account(m_mappings->WaitUntilAtMost(m_mappings->InFlight() - 1));   // may fill a slice and start its reduction
        return true;
    }
    return false;
}

bool HipSha256D::Instance::NewBatch(int dev)
{
    timing::Scope ts(timing::BATCH);
    for (;;) {
        m_batch = Dev(dev).batches->New();
        if (m_batch) return true;
        if (!m_ok || !WaitForMemory()) {
            std::cerr << "Failed to allocate a batch: " << vkmr_hip_last_error() << std::endl;
            return false;
        }
    }
}

// A launch of the map kernel fills the chip from about 2^19 strings on.  Batches are sized in bytes
// (64 MiB by default: best behind the host reader for short strings), so when the strings are long
// the following batches are made larger, up to batch_bytes_max.
void HipSha256D::Instance::AdaptBatchSize(const Batch& sent)
{
    if (sent.Count() == 0) return;
    const size_t avg = sent.Size() / sent.Count() + 4;
    if (avg < 128 + 4) return;   // short strings: a batch of the configured size already holds plenty
    size_t want = avg << 19;
    want = (want + ((size_t)64 << 20) - 1) & ~(((size_t)64 << 20) - 1);
    if (want > m_cfg.batch_bytes_max) want = m_cfg.batch_bytes_max;
    Batches& pool = *Dev(sent.Device()).batches;
    if (want > 4 * pool.DataBytes()) want = 4 * pool.DataBytes();   // by steps: the stream may be about to end
    if (want > pool.DataBytes() + pool.DataBytes() / 2) {
        if (m_cfg.verbose) std::cout << "Strings average " << avg - 4 << " bytes: batches of " << (want >> 20) << " MiB from now on." << std::endl;
        pool.Reshape(want, (size_t)1 << 20);
    }
}

bool HipSha256D::Instance::MapCurrent()
{
    Slice& slice = m_slices.Current();
    if (m_batch.Empty() || !slice) return true;
    if (m_mappings->InFlight() >= m_cfg.max_inflight) {
        timing::Scope tw(timing::MAP_WAIT);
        Account(m_mappings->WaitUntilAtMost(m_cfg.max_inflight - 1));
    }
    timing::Scope ts(timing::MAP);
    PerDevice& pd = Dev(slice.Device());
    AdaptBatchSize(m_batch);
    if (!pd.prefetched && !m_draining && m_batch.Words() * 4 >= pd.batches->DataBytes() / 2) {
        // the first batch of this device went out (about) full and the stream goes on: it will need the rest of the
        // pipeline's batches too -- pin them on a helper thread while this one keeps packing
        pd.prefetched = true;
        pd.batches->Prefetch(m_cfg.max_inflight);
    }
    if (m_devs.size() > 1 && !m_draining && m_batch.Words() * 4 >= pd.batches->DataBytes() / 2) {
        // several devices: the slice after this one goes to the next device of the deal -- its batches are pinned while
        // this slice fills, not when its first strings arrive (pinning stalls the packer: profiles/r03_frontend_setup_orders.txt)
        PerDevice& next = m_devs[((size_t)(&pd - m_devs.data()) + 1) % m_devs.size()];
        if (!next.prefetched) {
            next.prefetched = true;
            next.batches->Prefetch(m_cfg.max_inflight + 1);
        }
    }
    return m_mappings->Map(std::move(m_batch), slice.Sub(), pd.map_stream, pd.copy_stream) == VKMR_OK;
}

bool HipSha256D::Instance::StartSliceAndBatch()
{
    // no memory for the next slice (HBM full, or the slice budget used up): wait for the oldest
    // reduction to retire and take over its slice, instead of halting (reference SHA-256vk.cpp:396-399
    // halts; README.md:113 is the to-do this implements)
    for (;;) {
        timing::Scope ts(timing::SLICE);
        bool budget_hit = false;
        Slice& slice = m_slices.New(&budget_hit);
        if (slice) break;
        if (!m_ok || !WaitForMemory()) {
            std::cerr << "Failed to allocate slice " << m_slices.LastNumber() + 1 << ": "
                      << (budget_hit ? "slice budget used up and nothing in flight" : vkmr_hip_last_error()) << std::endl;
            return false;
        }
    }
    return NewBatch(m_slices.Current().Device());
}

bool HipSha256D::Instance::Add(const char* bytes, size_t size)
{
    if (!m_ok || !EnsureGeometry(nullptr, 0)) return false;
    // progress of in-flight work is discovered here, on the caller's thread
    // (reference SHA-256vk.cpp:318-335)
    m_reductions->Update();
    if (m_mappings->InFlight()) Account(m_mappings->Update());
    if (!m_ok || !m_reductions->Ok()) return (m_ok = false);   // a mapping or reduction failed on the device: stop reading

    // A failed allocation refuses this string but keeps what was added: the caller stops
    // reading and Root() still covers the earlier strings (reference: Add returns false,
    // the partial root is printed, src/vkmr/Vkmr.cpp:44-55, SHA-256vk.cpp:396-399).
    if (!m_slices.Current()) {
        if (!StartSliceAndBatch()) return false;
    } else if (m_slices.Current().Available() == 0) {
        // the slice is fully reserved: send off what is batched and open the next slice
        if (!MapCurrent()) return (m_ok = false);
        if (!StartSliceAndBatch()) return false;
    }
    if (!m_batch.Push(bytes, size)) {
        // batch full: map it and continue in a fresh one on the same device
        const int dev = m_slices.Current().Device();
        const bool was_empty = m_batch.Empty();
        if (!MapCurrent()) return (m_ok = false);
        if (was_empty && !GrowBatchesFor(dev, size)) {
            std::cerr << "A string of " << size << " byte(s) does not fit an empty batch." << std::endl;
            return false;
        }
        if (!NewBatch(dev) || !m_batch.Push(bytes, size)) {
            // refuse this string only: what was added before still has a root
            // (the caller stops reading and prints it, reference Vkmr.cpp:44-55)
            std::cerr << "A string of " << size << " byte(s) does not fit an empty batch." << std::endl;
            return false;
        }
    }
    return m_slices.Current().Reserve(1);
}

#ifdef VKMR_EXPERIMENTS
// The device-split path (HipConfig::device_split).  The host's part is one pass: the span's bytes into the batch's pinned
// text area (fork-join, equal byte ranges -- no line boundaries needed) with its newlines counted, so that the slice and
// the caller's tally know how many strings there are before the device has seen a byte.
size_t HipSha256D::Instance::PushTextForDevice(const char* text, size_t len, bool final, Tally* tally)
{
    Slice& slice = m_slices.Current();
    if (!m_batch.CanHoldText() || !m_batch.Empty() || !slice || len < ((size_t)1 << 20)) return 0;
    // the packed strings are a little larger than the text (padding to words): 7/8 of the batch's data area is taken as text,
    // which leaves room for lines of 25 bytes on average (shorter ones: the host packer); a last line without its newline gets one
    const size_t room = m_batch.TextCapacity() - m_batch.TextCapacity() / 8 - 1;
    size_t take = len < room ? len : room;
    bool add_newline = false;
    if (take < len || !final) {   // whole lines only
        const void* nl = memrchr(text, '\n', take);
        if (!nl) return 0;
        take = (size_t)(static_cast<const char*>(nl) - text) + 1;
    } else if (text[take - 1] != '\n') {
        add_newline = true;
    }
    if (take < ((size_t)1 << 20)) return 0;
    const unsigned threads = m_pool->Width();
    std::vector<TextCount> counts(threads, TextCount{0, 0});
    const uint8_t* src = reinterpret_cast<const uint8_t*>(text);
    uint8_t* dst = m_batch.TextArea();
    const size_t per = (((take + threads - 1) / threads) + 63) & ~(size_t)63;   // threads * per >= take
    {
        timing::Scope ts(timing::PACK);
        m_pool->Run(threads, [&](unsigned t) {
            const size_t lo = (size_t)t * per < take ? (size_t)t * per : take, hi = lo + per < take ? lo + per : take;
            if (hi > lo) counts[t] = CopyAndCountLines(src + lo, hi - lo, dst + lo, lo == 0 || src[lo - 1] == '\n');   // the span starts at a line's start
        });
    }
    uint64_t newlines = 0, empties = 0;
    for (const auto& c : counts) { newlines += c.newlines; empties += c.empties; }
    size_t text_bytes = take;
    if (add_newline) {
        dst[text_bytes++] = '\n';
        ++newlines;   // it ends a line that is not empty (the byte before it is not a newline)
    }
    const uint64_t strings = newlines - empties, payload = text_bytes - newlines;
    if (strings > slice.Available() || strings > m_batch.CapacityCount() || (payload + 3 * strings) / 4 > m_batch.CapacityWords())
        return 0;   // the slice's last strings, or an unusual text: the host packer cuts where it must
    tally->items += strings;
    tally->bytes += payload;
    tally->empties += empties;
    if (strings > 0) {
        m_batch.SetText(text_bytes, (size_t)strings, (size_t)payload);
        slice.Reserve((size_t)strings);
    }
    return take;
}
#endif   // VKMR_EXPERIMENTS

// Bulk ingest: whole spans of lines are packed straight into the pinned batch
// (memchr + memcpy per line, no per-line call or temporary), with the same batch/slice
// hand-offs as Add().
bool HipSha256D::Instance::AddLines(const char* buf, size_t len, bool final, Tally* tally)
{
    if (!m_ok || !EnsureGeometry(buf, len)) return false;
    size_t pos = 0;
    while (pos < len) {
        {
            timing::Scope ts(timing::UPDATE);
            m_reductions->Update();
            if (m_mappings->InFlight()) Account(m_mappings->Update());
        }
        if (!m_ok || !m_reductions->Ok()) return (m_ok = false);   // a mapping or reduction failed on the device: stop reading
        if (!m_slices.Current()) {
            if (!StartSliceAndBatch()) return false;
        } else if (m_slices.Current().Available() == 0) {
            if (!MapCurrent()) return (m_ok = false);
            if (!StartSliceAndBatch()) return false;
        }
        if (!m_batch.Empty() && len - pos >= (1u << 20)) {
            // A large span is waiting.  A batch that is full but for a sliver goes out as it is (topping it up line by line
            // would be done on this thread while the packer's other threads wait); so does a batch that holds a good part of
            // its capacity when the rest of the span will not fit: whole spans pack with one fork-join pair, cut ones with
            // two, and a default batch is made to hold one span of stdin (HipConfig::batch_bytes).
            const size_t room = m_batch.RoomWords(), cap = m_batch.CapacityWords();
            const double needs = (double)(len - pos) * (m_words_per_byte > 0.0 ? m_words_per_byte : 0.26) * 1.02;
            if (room < cap / 32 || (needs > (double)room && cap - room >= cap / 4)) {
                const int dev = m_slices.Current().Device();
                if (!MapCurrent() || !NewBatch(dev)) return (m_ok = false);
            }
        }
#ifdef VKMR_EXPERIMENTS
        if (m_cfg.device_split && len - pos >= ((size_t)1 << 20)) {
            // text for the device to split: into an empty batch (what the host packer left in the current one goes out first)
            if (!m_batch.Empty()) {
                const int dev = m_slices.Current().Device();
                if (!MapCurrent() || !NewBatch(dev)) return (m_ok = false);
            }
            const size_t took = PushTextForDevice(buf + pos, len - pos, final, tally);
            if (took) {
                pos += took;
                if (!m_batch.Empty()) {
                    const int dev = m_slices.Current().Device();
                    if (!MapCurrent() || !NewBatch(dev)) return (m_ok = false);
                }
                continue;
            }
        }
#endif
        Slice& slice = m_slices.Current();
        PackResult r = m_batch.PushLinesParallel(buf + pos, len - pos, final, slice.Available(), *m_pool, m_words_per_byte);
        if (r.consumed >= (1u << 20)) m_words_per_byte = (double)r.words / (double)r.consumed;
        if (r.consumed == 0) {
            timing::Scope ts(timing::PACK_SERIAL);
            r = m_batch.PushLines(buf + pos, len - pos, final, slice.Available());
        }
        slice.Reserve(r.strings);
        tally->items += r.strings;
        tally->bytes += r.bytes;
        tally->empties += r.empties;
        pos += r.consumed;
        if (pos >= len) break;
        if (slice.Available() == 0) continue;            // slice full: the loop head opens the next one
        if (r.strings == 0 && r.consumed == 0) {
            if (!final && !memchr(buf + pos, '\n', len - pos)) break;   // only an incomplete line is left
            // the batch is full (or the next line does not fit): map it, continue in a fresh one
            const int dev = slice.Device();
            const bool was_empty = m_batch.Empty();
            if (!MapCurrent()) return (m_ok = false);
            if (was_empty) {   // even an empty batch cannot take the next line: larger batches, if allowed
                const char* nl = static_cast<const char*>(memchr(buf + pos, '\n', len - pos));
                const size_t size = nl ? (size_t)(nl - (buf + pos)) : len - pos;
                if (!GrowBatchesFor(dev, size)) {
                    std::cerr << "A string of " << size << " byte(s) does not fit an empty batch." << std::endl;
                    return false;
                }
            }
            if (!NewBatch(dev)) return (m_ok = false);
        }
    }
    return true;
}

// A string that does not fit an empty batch of the current size: batches become large enough for it,
// within batch_bytes_max (the reference's batches are 256 MiB, SHA-256vk.cpp:23, :247-248; ours start
// smaller, so a string the reference would take must not be refused here).
bool HipSha256D::Instance::GrowBatchesFor(int dev, size_t string_bytes)
{
    size_t want = (string_bytes + 4 + ((size_t)64 << 20) - 1) & ~(((size_t)64 << 20) - 1);
    if (want > m_cfg.batch_bytes_max) want = m_cfg.batch_bytes_max;
    Batches& pool = *Dev(dev).batches;
    if (want <= pool.DataBytes() || (string_bytes + 3) / 4 * 4 > want) return false;
    if (m_cfg.verbose) std::cout << "A string of " << string_bytes << " bytes: batches of " << (want >> 20) << " MiB from now on." << std::endl;
    pool.Reshape(want, pool.MetaCount());
    return true;
}

bool HipSha256D::Instance::StagePacked(const uint32_t* data, const vkmr_metadata* meta, size_t count)
{
    if (!m_ok || !EnsureGeometry(nullptr, 0)) return false;
    auto stage_current = [&] {
        Slice& slice = m_slices.Current();
        if (m_batch.Empty() || !slice) return;
        Staged st;
        st.dev = slice.Device();
        st.sub = slice.Sub();
        st.batch = std::move(m_batch);
        m_staged.push_back(std::move(st));
    };
    // a failure gives up what was staged: the batches go back to their pools while the pools are alive
    auto give_up = [&] { m_staged.clear(); return false; };
    size_t pos = 0;
    while (pos < count) {
        if (!m_slices.Current()) {
            if (!StartSliceAndBatch()) return give_up();
        } else if (m_slices.Current().Available() == 0) {
            stage_current();
            if (!StartSliceAndBatch()) return give_up();
        }
        Slice& slice = m_slices.Current();
        const size_t took = m_batch.PushPacked(data, meta + pos, count - pos, slice.Available());
        slice.Reserve(took);
        pos += took;
        if (pos >= count) break;
        if (slice.Available() == 0) continue;
        if (took == 0) {   // the batch is full: hold it, continue in a fresh one of the same device
            const int dev = slice.Device();
            if (m_batch.Empty()) {
                std::cerr << "A packed string does not fit an empty batch." << std::endl;
                return give_up();
            }
            stage_current();
            if (!NewBatch(dev)) { m_ok = false; return give_up(); }
        }
    }
    return true;
}

ISha256D::out_type HipSha256D::Instance::RootOfStaged()
{
    if (!m_ok || !m_geometry) return "";
    {   // the residual batch joins the staged ones
        Slice& slice = m_slices.Current();
        if (!m_batch.Empty() && slice) {
            Staged st;
            st.dev = slice.Device();
            st.sub = slice.Sub();
            st.batch = std::move(m_batch);
            m_staged.push_back(std::move(st));
        }
    }
    for (auto& st : m_staged) {
        m_reductions->Update();
        if (m_mappings->InFlight()) Account(m_mappings->Update());
        if (m_mappings->InFlight() >= m_cfg.max_inflight) Account(m_mappings->WaitUntilAtMost(m_cfg.max_inflight - 1));
        PerDevice& pd = Dev(st.dev);
        if (!m_ok || m_mappings->Map(std::move(st.batch), std::move(st.sub), pd.map_stream, pd.copy_stream) != VKMR_OK) {
            m_staged.clear();   // nothing staged outlives the failure (the pools it points into go with the instance)
            return "";
        }
    }
    m_staged.clear();
    return Root();
}

ISha256D::out_type HipSha256D::Instance::Root()
{
    if (!m_ok || !m_geometry) return "";   // nothing was ever added: no root (reference CpuSha256D::Root on empty, SHA-256plus.cpp:494-496)
    // residual batch, then every mapping (reference SHA-256vk.cpp:291-299)
    m_draining = true;
    const bool single = m_slices.LastNumber() <= 1;
    {
        timing::Scope ts(timing::DRAIN_MAP);
        if (!MapCurrent()) return "";
        Account(m_mappings->WaitFor());
    }
    if (!m_ok) return "";   // a mapping or a reduction failed: there is no root
    timing::Scope ts(timing::DRAIN_REDUCE);
    // every remaining slice (at most the last, partial one): slice #1 alone is
    // reduced over its own count, any other to full capacity height
    // (reference Reductions.cpp:471; SHA-256vk.cpp:301-311)
    while (m_slices.Has()) {
        const Slice& any = m_slices.Any();
        const uint32_t number = any.Number();
        Slice s = m_slices.Remove(number);
        if (!s || s.Count() == 0) continue;
        const uint32_t height = (single && number == 1) ? tree_height(s.Count()) : m_cfg.slice_log2;
        const int dev = s.Device();
        if (m_reductions->Reduce(std::move(s), height, Dev(dev).reduce_stream) != VKMR_OK) return "";
    }
    const out_type root = m_reductions->WaitFor();
    if (m_cfg.verbose) {
        size_t batches = 0;
        for (const auto& pd : m_devs) batches += pd.batches->Allocations();
        std::cout << "Allocations: " << m_slices.Allocations() << " slice(s), " << batches << " batch(es), " << m_reductions->Allocations()
                  << " reduction scratch buffer(s) for " << m_slices.LastNumber() << " slice(s)." << std::endl;
    }
    return root;
}

}  // namespace vkmr
