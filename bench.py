#!/usr/bin/env python3
"""bench.py -- leaf hashes/s and Merkle-root wall time on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--leaves-log2 26] [--maxlen 127]

One "step" = one pass of the hot path (map -> reduce -> gather -> combine) over the whole
synthetic workload, with the packed batches already resident in HBM.  The workload is
BASELINE.json's configs[2]: `rndm <seed> 2^26 127` (the restated glibc rand() generator in
csrc/host/rndm_stream.cpp produces the identical strings straight into packed batches).

N > 1 is configs[3] (2^29 leaves on 8 GPUs): one process per GPU, rank r holds the 2^26 leaves
of `rndm 42+r` as ONE slice (north star: slices shard one-per-GPU), maps and reduces them to
one 32-byte sub-tree root, the N roots cross xGMI in ONE RCCL all-gather issued through the
C ABI (vkmr_hip_gather_roots_async) and are combined on the GPU.  No data-path collective
besides that; weak scaling.  The ranks are either started by torch.distributed.run (the driver's
form) or -- `python bench.py --gpus N` without WORLD_SIZE -- by this script itself, as N child
processes spawned before anything touches the GPU.  Fewer than N GPUs, or RCCL unusable: the
run FAILS; it never reports a smaller N or another transport as if it were the requested one.

Prints ONE JSON line on rank 0 (see the driver contract): `value` is whole-job leaf hashes/s;
`roofline` prices the dominant kernel (map) against HBM (the spec'd bound), `roofline_reduce`
the reduction, `valu_roofline` carries the bound the path actually sits under -- VALU issue: the
shader clock each kernel HOLDS, measured in-kernel in this run, and the dual-issue floor of its
instruction stream (static counts of the loaded library);
`root_matches_golden` checks the timed path's root (and every rank's sub-root) against
tests/golden/big_roots.json, which holds what the reference's own CPU path printed for these
streams; `cpu_baseline` is that CPU path (oracle/_ref, built from the reference sources) timed
on the SAME 2^26-string stream on this box's host cores (one core: the path is serial; ~80 s); `long_strings` is the map
kernel on `rndm 42 2^21 4096` (BASELINE configs[4]'s shape, one 4.3 GB batch).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0            # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)
SIMDS = 256 * 4                   # 256 CUs x 4 SIMDs; one VALU issue turn = 4 shader cycles
MAX_CLOCK_GHZ = 2.4


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--leaves-log2", type=int, default=26, help="leaves per GPU (default 2^26 = configs[2])")
    p.add_argument("--maxlen", type=int, default=127, help="rndm max string length argument")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--slice-log2", type=int, default=None,
                   help="digests per slice (default: leaves-log2, ONE slice per GPU; the reference's slice is 2^23 = 256 MiB)")
    p.add_argument("--batch-log2", type=int, default=None,
                   help="strings per map launch (default: leaves-log2, the GPU's whole share as ONE packed batch -- 4.4 GB, well inside "
                        "the format's 2^32 words; the reference's Vulkan-sized batches hold <= 2^23, which costs 3 %% in launch tails)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-log2", type=int, default=None,
                   help="give the CPU baseline only the first 2^k strings of the stream (default: the whole stream, about 80 s for 2^26)")
    p.add_argument("--no-clock-leg", action="store_true", help="skip the in-kernel shader-clock measurement (a child process on the stamped twin of the library)")
    p.add_argument("--levels-variant", action="store_true", help="use the one-level-per-launch reduction")
    p.add_argument("--rehearse-gloo", action="store_true",
                   help="N > 1 on a box with fewer GPUs: ranks share GPU 0 and the roots travel over gloo.  A plumbing "
                        "rehearsal, labelled as such in the output; never a scaling result")
    p.add_argument("--force-dist", action="store_true", help="form the process group and the RCCL communicator even with one rank")
    p.add_argument("--no-pipeline", action="store_true", help="skip the PCIe-inclusive (pinned host -> root) measurement")
    p.add_argument("--no-long-strings", action="store_true", help="skip the secondary long-string map measurement")
    p.add_argument("--no-config5", action="store_true", help="skip the configs[4] full-size leg (rndm 42 2^24 4096 through map + reduce: about a minute of host-side generation)")
    return p.parse_args()


# ---- N ranks without torch.distributed.run -------------------------------------------------------

def visible_gpus():
    """GPUs this job can use, counted in a short-lived child process: the process that spawns the ranks
    must never have touched the GPU itself."""
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    try:
        return int(r.stdout.decode().strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def launch_ranks(a):
    """`python bench.py --gpus N` with no WORLD_SIZE: start the N ranks here, one child process per GPU,
    before any GPU call in this process; pass rank 0's JSON line through; non-zero if any rank fails."""
    have = visible_gpus()
    if have < a.gpus and not a.rehearse_gloo:
        sys.stderr.write(f"[bench] --gpus {a.gpus} requested but this node has {have} GPU(s): refusing to run fewer ranks "
                         f"or to share a GPU (use --rehearse-gloo for a plumbing rehearsal)\n")
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:       # one rank failed: the others would wait for it until the timeout
                    q.terminate()
        time.sleep(0.05)
    return rc


# ---- CPU baseline --------------------------------------------------------------------------------

def cpu_baseline(seed, maxlen, sample_log2, full_log2):
    """The reference CPU-serial path (oracle/_ref/vkmr_cpu_ref: the reference's own SHA-256plus / Inputs / StopWatch
    sources, g++ -O2) on the same rndm stream through stdin, timed by its own stopwatch line, which spans reading,
    hashing and the tree (reference Vkmr.cpp:36-55).  By default the WHOLE stream (2^full_log2 strings); with
    --cpu-sample-log2 a stated prefix, never extrapolated.  Falls back to the C restatement ("port") when the
    reference build is absent."""
    n = 1 << sample_log2
    rndm = os.path.join(ROOT, "vk_merkle_roots_amd", "bin", "rndm")
    ref = os.path.join(ROOT, "oracle", "_ref", "vkmr_cpu_ref")
    whole = sample_log2 == full_log2
    sample = (f"the whole stream, rndm {seed} 2^{sample_log2} {maxlen}, via stdin" if whole
              else f"PREFIX: the first 2^{sample_log2} of the 2^{full_log2} strings of rndm {seed} * {maxlen}, via stdin")
    if os.path.exists(ref) and os.path.exists(rndm):
        gen = subprocess.Popen([rndm, str(seed), str(n), str(maxlen)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        out = subprocess.run([ref], stdin=gen.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        gen.wait()
        line = [l for l in out.splitlines() if "computed root" in l]
        if line:
            ms = float(line[0].rsplit(" in ", 1)[1])
            root = line[0].split("=> ")[1].split(" in ")[0]
            return {"value": n / (ms / 1e3), "unit": "leaf hashes/s", "cores": 1, "kind": "reference",
                    "sample": sample + " (stdin parse + hash + tree, the program's own stopwatch; g++ -O2)",
                    "host_cores": os.cpu_count(), "seconds": ms / 1e3, "leaves": n, "root": root}
    # port: the oracle library on a packed sample (checker timed as a baseline, never shipped)
    import vk_merkle_roots_amd as vk
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    b = vk.rndm_packed(seed, n, maxlen)
    leaves = np.zeros((n, 8), dtype=np.uint32)
    root = np.zeros(8, dtype=np.uint32)
    t0 = time.perf_counter()
    L.oracle_leaves_packed(C.c_void_p(b.data.ctypes.data), C.c_void_p(b.meta.ctypes.data), C.c_size_t(n),
                           C.c_void_p(leaves.ctypes.data), C.c_int(1))
    L.oracle_root_inplace(C.c_void_p(leaves.ctypes.data), C.c_size_t(n), C.c_void_p(root.ctypes.data))
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "leaf hashes/s", "cores": 1, "kind": "port",
            "sample": sample.replace("via stdin", "packed") + " (hash + tree only)", "host_cores": os.cpu_count(), "seconds": dt, "leaves": n}


# ---- secondary measurements (N = 1 only, outside the timed region) ---------------------------------

def pipeline_rate(vk, batch, bstr, slice_log2, device=0, runs=3):
    """Pipeline-level rate (SURVEY.md 8d ii): packed batches in PINNED host memory -> root, on the PRODUCT's schedule:
    libvkmr_pipeline.so stages the strings into the C++ stream processor's own pinned batches (untimed), then runs what
    `vkmr hip:<n>` runs -- per batch Mappings::Map (two H2D copies on the device's copy stream, the map kernel behind
    them on its map stream), slices to Reductions as they fill, the combine of the slice roots.  PCIe-inclusive;
    reported beside `value`, never as it."""
    from vk_merkle_roots_amd.build import PIPELINE_LIB
    L = C.CDLL(PIPELINE_LIB)
    L.vkmr_host_pipeline_packed.restype = C.c_int
    L.vkmr_host_pipeline_packed.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_char_p,
                                            C.POINTER(C.c_double)]
    best, root = None, None
    times = []
    for _ in range(runs + 1):          # the first run also pays for pinning the pool's buffers' first touch
        hexbuf, secs = C.create_string_buffer(65), C.c_double()
        rc = L.vkmr_host_pipeline_packed(device, batch.data.ctypes.data, batch.words, batch.meta.ctypes.data, batch.count, bstr, slice_log2,
                                         hexbuf, C.byref(secs))
        if rc != 0:
            raise RuntimeError("vkmr_host_pipeline_packed failed: " + vk._abi.what_error())
        times.append(secs.value)
        root = hexbuf.value.decode()
    dt = float(np.median(times[1:]))
    # what crosses the link: the packed words and, per string, its 16-bit size (the entries are written on the device:
    # vkmr_hip_metadata_from_sizes_async) -- or the 8-byte entry when a string reaches 65 535 bytes or VKMR_SEND_METADATA=1
    per_string = 8 if (os.environ.get("VKMR_SEND_METADATA", "0") not in ("", "0") or int(batch.meta[:, 1].max()) >= 65535) else 2
    return {"leaf_hashes_per_s": batch.count / dt, "ms": dt * 1e3, "h2d_GBps": (batch.words * 4 + batch.count * per_string) / dt / 1e9,
            "h2d_bytes_per_string_of_metadata": per_string, "root_hex": root, "runs_ms": [t * 1e3 for t in times]}


def from_file_rate(seed, count, maxlen, device, runs=3, timeout_s=300):
    """The reference's own measurement shape (src/vkmr/Vkmr.cpp:28-58: newline-separated strings on stdin, one stopwatch
    around reading, hashing and the tree): the product's `vkmr hip:<n>` fed the SAME stream as the timed steps from a
    file, its own "... in <ms>" line and the process wall clock.  Host packer + PCIe + kernels; reported beside `value`,
    never as it."""
    import tempfile
    rndm = os.path.join(ROOT, "vk_merkle_roots_amd", "bin", "rndm")
    vkmr = os.path.join(ROOT, "vk_merkle_roots_amd", "bin", "vkmr")
    if not (os.path.exists(rndm) and os.path.exists(vkmr)):
        return {"error": "vk_merkle_roots_amd/bin/{rndm,vkmr} missing: run `python -m vk_merkle_roots_amd.build`"}
    with tempfile.NamedTemporaryFile(prefix="vkmr_stream_", suffix=".txt") as tmp:
        subprocess.run([rndm, str(seed), str(count), str(maxlen)], stdout=tmp, stderr=subprocess.DEVNULL, check=True, timeout=timeout_s)
        tmp.flush()
        nbytes = os.path.getsize(tmp.name)
        printed, walls, root, items = [], [], None, None
        for _ in range(runs + 1):   # the first run also warms the page cache's mapping of the file
            with open(tmp.name, "rb") as f:
                t0 = time.perf_counter()
                r = subprocess.run([vkmr, f"hip:{device}"], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout_s)
                walls.append(time.perf_counter() - t0)
            line = [l for l in r.stdout.decode().splitlines() if "computed root" in l]
            if r.returncode != 0 or not line:
                return {"error": f"vkmr hip:{device} failed (rc {r.returncode})"}
            printed.append(float(line[-1].rsplit(" in ", 1)[1]))
            root = line[-1].split("=> ")[1].split(" in ")[0]
            items = int(line[-1].split("(of ")[1].split(" item")[0])
    ms = float(np.median(printed[1:]))
    return {"workload": f"rndm {seed} {count} {maxlen} in a file on stdin ({nbytes} bytes)", "printed_ms": ms, "printed_ms_runs": printed,
            "process_wall_s": float(np.median(walls[1:])), "leaf_hashes_per_s": count / (ms / 1e3), "text_GBps": nbytes / (ms / 1e3) / 1e9,
            "items": items, "root_hex": root,
            "what": "`vkmr hip:<n> < file`: mapped file -> 16-thread indexed packer -> pinned batches -> H2D on the copy stream, map kernel behind "
                    "it -> slices of 2^23 reduced as they fill -> combine; the program's own stopwatch (it starts, like the reference's, after "
                    "the backend is constructed)"}


def two_stream_rate(dev, vk, d_batch, bstr, n, slice_height, steps=8):
    """The same work with the map on one stream and the reduction on another, as the stream processor runs them
    (csrc/host/hip_sha256d.cpp: map_stream / reduce_stream per device): step k+1's map fills the chip while step k's
    reduction is in its latency-bound top (0.3 ms of mostly idle GPU), and the host does not wait per step.  Two
    digest buffers, events between the streams.  Reported beside `value`, which stays the one-stream figure."""
    d_data, words, d_meta = d_batch
    map_stream, red_stream = dev.new_stream(), dev.new_stream()
    bufs = [dev.alloc(32 * n) for _ in range(2)]
    scratch = [dev.alloc(dev.lib.vkmr_hip_reduce_scratch_bytes(n)) for _ in range(2)]
    d_roots = dev.alloc(32 * (steps + 1))
    mapped = [dev.new_event() for _ in range(2)]
    reduced = [dev.new_event() for _ in range(2)]

    def run(k_steps):
        for k in range(k_steps):
            z = k & 1
            if k >= 2:   # the digest buffer is free once the reduction that read it has finished
                vk.check(dev.lib.vkmr_hip_stream_wait_event(dev.index, map_stream, reduced[z]), "wait")
            dev.map_async(d_data, words, d_meta, bstr, bufs[z], stream=map_stream)
            dev.record(mapped[z], map_stream)
            vk.check(dev.lib.vkmr_hip_stream_wait_event(dev.index, red_stream, mapped[z]), "wait")
            vk.check(dev.lib.vkmr_hip_reduce_async(dev.index, red_stream, bufs[z].ptr, n, slice_height, scratch[z].ptr, d_roots.at(32 * k)), "reduce")
            dev.record(reduced[z], red_stream)
        dev.sync(map_stream)
        dev.sync(red_stream)

    run(2)
    t0 = time.perf_counter()
    run(steps)
    dt = (time.perf_counter() - t0) / steps
    roots = dev.download(d_roots, 32 * steps).reshape(-1, 8)
    for b in bufs + scratch + [d_roots]:
        b.free()
    return {"leaf_hashes_per_s": n / dt, "ms_per_step": dt * 1e3, "roots": roots,
            "what": "map on one stream, reduction on another, two digest buffers, no host wait per step"}


def long_strings_rate(dev, vk, seed, count_log2=21, maxlen=4096, launches=10):
    """BASELINE configs[4]'s shape on the map kernel: rndm <seed> 2^21 4096 (lengths 1..4095, 1..65 blocks per
    string) as ONE batch of about 4.3 GB -- the size a long-string batch needs to fill the chip."""
    n = 1 << count_log2
    b = vk.rndm_packed(seed, n, maxlen)
    d_data, d_meta, d_out = dev.upload(b.data), dev.upload(b.meta), dev.alloc(32 * n)
    ev = [(dev.new_event(), dev.new_event()) for _ in range(launches)]
    # the GPU has idled for the seconds the host took to generate this batch: warm up until it is back at its steady
    # clocks (the first tens of milliseconds after idle run 10 % slower: profiles/r02_clock_power.txt)
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.4:
        for _ in range(4):
            dev.map_async(d_data, b.words, d_meta, n, d_out)
        dev.sync()
    for e0, e1 in ev:
        dev.record(e0)
        dev.map_async(d_data, b.words, d_meta, n, d_out)
        dev.record(e1)
    dev.sync()
    ms = float(np.mean([dev.elapsed_ms(e0, e1) for e0, e1 in ev]))
    sizes = b.meta[:, 1].astype(np.int64)
    blocks = int(((sizes + 8) // 64 + 1).sum())
    nbytes = b.words * 4 + 40 * n
    info = dev.lib.vkmr_hip_kernel_info().decode()
    for buf in (d_data, d_meta, d_out):
        buf.free()
    # HBM bytes per launch from the PMC passes over the same workload AND the same build of the same kernel
    from vk_merkle_roots_amd import provenance
    rec = provenance.load_pmc()
    traffic, tsrc = provenance.traffic_from_pmc(dict(rec or {}, map_kernel_symbol=(rec or {}).get("long_strings_map_kernel_symbol")), info,
                                                "long_strings_map_hbm_bytes_per_launch", long_strings_workload=f"rndm {seed} 2^{count_log2} {maxlen}, one batch")
    return {"workload": f"rndm {seed} 2^{count_log2} {maxlen}, one batch", "strings": n, "input_bytes": int(b.words * 4), "map_ms": ms,
            "map_mode": info.split(" reduce=")[0],
            "leaf_hashes_per_s": n / (ms * 1e-3), "roofline": {"bound": "hbm", "kernel": "map_kernel (" + (provenance.map_symbol_of(info) or "?") + ")", "achieved": nbytes / (ms * 1e-3) / 1e9,
                                                               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                               "traffic": traffic, "traffic_source": tsrc, "algorithmic_bytes_per_launch": int(nbytes)},
            "blocks": blocks, "compressions_per_string": blocks / n + 1}


def config5_full(dev, vk, seed, count_log2=24, maxlen=4096, batch_log2=20, slice_log2=23):
    """BASELINE configs[4] at full size through the whole path: `rndm <seed> 2^24 4096` (34 GB of text, 1..65 blocks per string)
    streamed as 16 packed batches of 2^20 strings (a packed batch addresses at most 2^32 words) -- each generated on the host,
    uploaded, then mapped with the input RESIDENT in HBM (the timed part: HIP events around every map launch) -- into 2
    slices of 2^23 digests (the reference's slice), one batched reduction of the slices, the combine of the two roots.
    `value` stays configs[2]'s figure; this block is the measured record of configs[4] (VERDICT r3 #2 / Missing #3)."""
    from vk_merkle_roots_amd.engine import digest_hex, tree_height
    from vk_merkle_roots_amd import provenance
    n, per, cap = 1 << count_log2, 1 << batch_log2, 1 << slice_log2
    nslices = n // cap
    stream = vk.RndmStream(seed, maxlen)
    d_digests = dev.alloc(32 * n)
    d_scratch = dev.alloc(dev.lib.vkmr_hip_reduce_slices_scratch_bytes(cap, nslices))
    d_roots, d_top, d_final = dev.alloc(32 * nslices), dev.reduce_scratch(max(nslices, 2)), dev.alloc(32)
    e0, e1 = dev.new_event(), dev.new_event()
    map_ms, words_total, blocks, payload = [], 0, 0, 0
    t_host = time.perf_counter()
    info = None
    for b0 in range(0, n, per):
        batch = stream.next(per)
        d_data, d_meta = dev.upload(batch.data), dev.upload(batch.meta)
        # The GPU idled through seconds of generation and upload and its clock ramps for tens of milliseconds afterwards (a cold
        # launch of this batch reads 1.42 ms, a settled one 1.14: profiles/r04_long_strings_tile_sweep.txt): 60 ms of untimed
        # launches of this batch first -- as the headline's settling steps -- then the median of three timed ones.
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < 0.06:
            for _ in range(4):
                dev.map_async(d_data, batch.words, d_meta, per, d_digests, out_offset_digests=b0)
            dev.sync()
        three = []
        for _ in range(3):
            dev.record(e0)
            dev.map_async(d_data, batch.words, d_meta, per, d_digests, out_offset_digests=b0)
            dev.record(e1)
            dev.sync()
            three.append(dev.elapsed_ms(e0, e1))
        map_ms.append(float(np.median(three)))
        info = info or dev.lib.vkmr_hip_kernel_info().decode()
        words_total += batch.words
        sizes = batch.meta[:, 1].astype(np.int64)
        blocks += int(((sizes + 8) // 64 + 1).sum())
        payload += int(sizes.sum())
        d_data.free()
        d_meta.free()
    stream.close()
    host_s = time.perf_counter() - t_host

    def reduce_all():
        dev.reduce_slices_async(d_digests, nslices, cap, cap, slice_log2, d_scratch, d_roots)
        dev.reduce_async(d_roots, nslices, tree_height(nslices), d_top, d_final)
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.06:
        reduce_all()
        dev.sync()
    red_ms = []
    for _ in range(5):
        dev.record(e0)
        reduce_all()
        dev.record(e1)
        dev.sync()
        red_ms.append(dev.elapsed_ms(e0, e1))
    root = digest_hex(dev.download(d_final, 32))
    for b in (d_digests, d_scratch, d_roots, d_top, d_final):
        b.free()
    total_map, red = float(np.sum(map_ms)), float(np.median(red_ms))
    golden = None
    try:
        rec = json.load(open(os.path.join(ROOT, "tests", "golden", "big_roots.json"))).get("config5")
        if rec and rec.get("generator") == f"rndm {seed} {n} {maxlen}":
            golden = rec
    except (OSError, ValueError):
        pass
    nbytes = words_total * 4 + 40 * n      # packed words + 8 B entry read, 32 B digest written, per string
    pmc = provenance.load_pmc()
    traffic, tsrc = provenance.traffic_from_pmc(dict(pmc or {}, map_kernel_symbol=(pmc or {}).get("long_strings_map_kernel_symbol")), info,
                                                "long_strings_map_hbm_bytes_per_launch", long_strings_workload=f"rndm {seed} 2^21 {maxlen}, one batch")
    # the counter record is of ONE batch of 2^21 strings of the same distribution and the same kernel: scaled by the strings here
    if traffic is not None:
        traffic = traffic * n / float(1 << 21)
        tsrc = dict(tsrc or {}, scaled="per-launch record of 2^21 strings of the same distribution x 8 (2^24 strings)")
    return {"workload": f"rndm {seed} 2^{count_log2} {maxlen} (BASELINE configs[4]), {n // per} packed batches of 2^{batch_log2} strings, {nslices} slices of 2^{slice_log2}",
            "strings": n, "payload_bytes": payload, "packed_bytes": int(words_total * 4), "blocks": blocks, "compressions_per_string": blocks / n + 1,
            "map_ms_total": total_map, "map_ms_per_batch": [round(x, 4) for x in map_ms], "reduce_and_combine_ms": red,
            "ms_total": total_map + red, "leaf_hashes_per_s": n / ((total_map + red) * 1e-3), "input_GBps": words_total * 4 / (total_map * 1e-3) / 1e9,
            "map_mode": (info or "").split(" reduce=")[0], "root": root,
            "root_matches": (root == golden["root"]) if golden else None,
            "golden": "tests/golden/big_roots.json config5 (the reference's CPU path on the same stream)" if golden else None,
            "roofline": {"bound": "hbm", "kernel": "map_kernel (" + (provenance.map_symbol_of(info or "") or "?") + ")", "achieved": nbytes / (total_map * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": nbytes / (total_map * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": tsrc, "algorithmic_bytes": int(nbytes)},
            "host_seconds_generating_and_uploading": host_s,
            "what": "resident-input kernel time of all 16 map launches (each: median of three after 60 ms of untimed launches of that batch, so that the clock "
                    "has settled after the host-side generation) + one batched reduction of 2 slices + combine; generation and upload of each batch are outside the timed region"}


def hip_all_check(timeout_s=120):
    """On a multi-GPU node: the C++ front end's one-process path, `rndm 42 2^20 127 | vkmr hip:all` with slices of
    2^17 -- eight slices dealt over the node's GPUs, ONE RCCL all-gather of their roots inside that process
    (vkmr_hip_comm_init_all + vkmr_hip_gather_roots_async), combine on GPU 0 -- against the golden root of that stream
    (tests/golden/vectors.json, printed by the reference's CPU path).  A correctness probe run by rank 0 after the
    timed region; a failure is reported in the field, it does not change `value`."""
    vkmr = os.path.join(ROOT, "vk_merkle_roots_amd", "bin", "vkmr")
    rndm = os.path.join(ROOT, "vk_merkle_roots_amd", "bin", "rndm")
    try:
        want = json.load(open(os.path.join(ROOT, "tests", "golden", "vectors.json")))["streams"]["G3_rndm_42_1048576_127"]["root"]
        import tempfile
        with tempfile.NamedTemporaryFile(prefix="vkmr_g3_", suffix=".txt") as tmp:   # a file, not a pipe: nothing can block on a full pipe
            subprocess.run([rndm, "42", "1048576", "127"], stdout=tmp, stderr=subprocess.DEVNULL, check=True, timeout=timeout_s)
            tmp.flush()
            env = dict(os.environ, VKMR_SLICE_LOG2="17")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "VKMR_HIP_LIB"):
                env.pop(k, None)
            t0 = time.perf_counter()
            with open(tmp.name, "rb") as f:
                child = subprocess.Popen([vkmr, "hip:all"], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
                try:
                    so, se = child.communicate(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    child.kill()
                    child.communicate()
                    return {"ran": True, "root_matches_golden": False, "error": f"no answer within {timeout_s} s"}
        line = [l for l in so.decode().splitlines() if "computed root" in l]
        if child.returncode != 0 or not line:
            return {"ran": child.returncode == 0, "root_matches_golden": None if b"No device selected" in se else False,
                    "error": (se.decode().strip().splitlines() or ["no output"])[-1][:200]}
        root = line[-1].split("=> ")[1].split(" in ")[0]
        return {"ran": True, "what": "rndm 42 2^20 127 | vkmr hip:all, 8 slices of 2^17 dealt over the node's GPUs, one RCCL all-gather in one process",
                "root_matches_golden": root == want, "root": root, "seconds": time.perf_counter() - t0}
    except Exception as e:   # a probe must not take the bench line down with it
        return {"ran": False, "root_matches_golden": None, "error": repr(e)[:200]}


def static_counts():
    """VALU instruction counts of the hash blocks of the library actually loaded (written beside it by the build from the
    assembly that was assembled into it: vk_merkle_roots_amd/isa_prio_pass.py: hash_blocks)."""
    from vk_merkle_roots_amd.build import HIP_LIB
    path = os.path.splitext(os.environ.get("VKMR_HIP_LIB", HIP_LIB))[0] + ".isa.json"
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    hb = rec.get("hash_blocks", {})
    node = next((v[0] for k, v in hb.items() if "reduce_pass_kernel" in k and v), None)
    out = {"build": rec.get("build"), "file": os.path.relpath(path, ROOT), "node": node, "map": {},
           "issue_pass": {"prio_gap": rec.get("prio_gap"), "split_add3_every": rec.get("split_add3_every"),
                          "complex_runs_raised": (rec.get("pass") or {}).get("runs"), "add3_split": (rec.get("pass") or {}).get("add3_split")}}
    for k, v in hb.items():
        if "map_kernel" in k and len(v) >= 2:
            m = k.split("map_kernelILi")[1].split("EEv")[0].replace("ELi", ", ").replace("ELb", ", ")
            out["map"][m] = {"block": v[0], "digest": v[-1]}     # the digest hash is the last hash block of a map kernel (the two-blocks-per-trip mode has two block bodies)
    return out


def clock_leg(seed, maxlen, leaves_log2):
    """The shader clock map_kernel and reduce_pass_kernel HOLD, measured in-kernel (s_memtime against the constant-rate
    s_memrealtime, per workgroup, after 2 s of back-to-back launches) by tools/kernel_clock.py in a child process on
    the stamped twin of the library (same source, -DVKMR_STAMPS; in the product no stamp executes).  Same workload SIZE as the
    timed steps: the clock a 5 ms launch holds is not the one a 1.3 ms launch holds (profiles/r04_kernel_clock_fault.txt)."""
    tool = os.path.join(ROOT, "tools", "kernel_clock.py")
    try:
        r = subprocess.run([sys.executable, tool, "--leaves-log2", str(leaves_log2), "--maxlen", str(maxlen), "--seed", str(seed)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400, env={k: v for k, v in os.environ.items() if k != "VKMR_HIP_LIB"})
        rec = json.loads(r.stdout.decode().strip().splitlines()[-1])
    except Exception as e:   # the leg must not take the bench line down with it
        return {"error": repr(e)[:200]}
    return {"method": rec.get("method"), "workload": rec.get("workload"), "library": rec.get("library"), "max_GHz": MAX_CLOCK_GHZ,
            "power_cap_W": rec.get("power_cap_W"),
            "map_kernel": {k: rec["map_kernel"].get(k) for k in ("GHz_median", "GHz_p5", "GHz_p95", "board_power_W", "sclk_sysfs_MHz", "phase_share")},
            "reduce_pass_kernel": {k: rec["reduce_pass_kernel"].get(k) for k in ("GHz_median", "GHz_p5", "GHz_p95", "board_power_W", "sclk_sysfs_MHz")}}


def issue_block(units, valu_per_unit, turns_per_unit, ms, clock_ghz):
    """One kernel against the dual-issue floor: `units` hashes (leaves or nodes) spread over all SIMDs, 64 per wavefront
    instruction; a turn is 4 shader cycles and takes at most two VALU instructions, one of them a simple one."""
    per_simd = units / 64.0 / SIMDS
    out = {"units": int(units), "valu_instr_per_unit": valu_per_unit, "floor_turns_per_unit": turns_per_unit, "ms": ms,
           "floor_ms_at_max_clock": per_simd * turns_per_unit * 4 / (MAX_CLOCK_GHZ * 1e6)}
    if clock_ghz:
        cycles = ms * clock_ghz * 1e6
        out.update(clock_GHz=clock_ghz, cycles_per_valu_instr=cycles / (per_simd * valu_per_unit),
                   floor_cycles_per_valu_instr=4.0 * turns_per_unit / valu_per_unit,
                   floor_ms_at_measured_clock=per_simd * turns_per_unit * 4 / (clock_ghz * 1e6),
                   frac_of_floor=per_simd * turns_per_unit * 4 / cycles)
    return out


def gather_label(distributed, through_abi):
    """What carries the sub-tree roots between ranks, as the JSON line names it."""
    if not distributed:
        return None
    return ("vkmr_hip_gather_roots_async: 1 ncclAllGather of 32 B per rank (C ABI, librccl)" if through_abi
            else "torch.distributed gather over gloo (rehearsal)")


def workload_label(seed, leaves_log2, maxlen, world):
    return (f"rndm {seed}+rank 2^{leaves_log2} {maxlen} per GPU (BASELINE configs[{2 if world == 1 else 3}]"
            f"{'' if world in (1, 8) else ' shape at ' + str(world) + ' GPUs'})")


def check_against_golden(golden, seed, world, root_hex, sub_roots):
    """(root_matches_golden, sub_roots_match_golden): the timed path's root and every rank's sub-tree root against what the
    reference's CPU path printed for these streams (tests/golden/big_roots.json: sub-roots of seeds 42..49, and their
    combination in rank order for 1..8 ranks); None where the file has no entry."""
    if not golden:
        return None, None
    want = golden.get("combined", {}).get(str(world))
    root_ok = (root_hex == want) if want else None
    have = [golden["sub_roots"].get(str(seed + r), {}).get("root") for r in range(world)]
    subs_ok = (sub_roots == have) if all(have) else None
    return root_ok, subs_ok


def golden_big_roots(leaves_log2, maxlen):
    path = os.path.join(ROOT, "tests", "golden", "big_roots.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    if rec.get("count") != (1 << leaves_log2) or rec.get("maxlen") != maxlen:
        return None
    return rec


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner and gloo a connection
    # note to file descriptor 1, so everything but the result line goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the rank count must be the one asked for")
    dist = None
    tdev = None
    collective = None
    if world > 1 or a.force_dist:
        import datetime
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if a.rehearse_gloo:
            collective = "gloo (REHEARSAL: ranks share GPU 0, roots over host memory -- not RCCL, not a scaling result)"
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
            local_rank = 0
        else:
            if torch.cuda.device_count() <= local_rank:
                isolated = any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
                if not (isolated and torch.cuda.device_count() == 1):
                    raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible")
                local_rank = 0   # the launcher gave this rank its own GPU through *_VISIBLE_DEVICES (RCCL refuses two ranks on one GPU)
            collective = "nccl"   # torch.distributed's nccl backend IS RCCL on ROCm; any failure here ends the run
            torch.cuda.set_device(local_rank)
            tdev = torch.device("cuda", local_rank)
            dist.init_process_group(backend="nccl", device_id=tdev, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=300))
            probe = torch.ones(1, device=tdev)
            dist.all_reduce(probe)            # first RCCL collective of the control plane: fail here, not inside the timed region
            torch.cuda.synchronize()
            assert int(probe.item()) == world

    import vk_merkle_roots_amd as vk
    from vk_merkle_roots_amd.engine import digest_hex, tree_height

    dev = vk.HipDevice(local_rank)
    n = 1 << a.leaves_log2
    slice_log2 = a.leaves_log2 if a.slice_log2 is None else min(a.slice_log2, a.leaves_log2)
    cap = 1 << slice_log2
    bstr = 1 << (a.leaves_log2 if a.batch_log2 is None else min(a.batch_log2, a.leaves_log2))
    nslices = n // cap
    nbatches = n // bstr

    # ---- the data-path communicator: RCCL through the C ABI, one rank per process ------------------
    comm = None
    if dist is not None and not a.rehearse_gloo:
        import torch
        uid = torch.zeros(vk._abi.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            buf = C.create_string_buffer(vk._abi.COMM_ID_BYTES)
            vk.check(dev.lib.vkmr_hip_comm_create_id(buf), "vkmr_hip_comm_create_id")
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        uid = uid.to(tdev)
        dist.broadcast(uid, src=0)            # the 128-byte id travels over the control plane
        uid_bytes = bytes(uid.cpu().numpy().tobytes())
        comm = C.c_void_p()
        vk.check(dev.lib.vkmr_hip_comm_init_rank(local_rank, uid_bytes, world, rank, C.byref(comm)), "vkmr_hip_comm_init_rank")

    # ---- synthetic input: rndm stream of this rank, packed, resident in HBM -----------
    t0 = time.perf_counter()
    batch = vk.rndm_packed(a.seed + rank, n, a.maxlen)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    d_batches = []          # one device-resident packed batch per map launch (own buffers, metadata rebased)
    for b in range(nbatches):
        sub = batch.slice(b * bstr, (b + 1) * bstr)
        d_batches.append((dev.upload(sub.data), sub.words, dev.upload(sub.meta)))
    t_h2d = time.perf_counter() - t0
    input_bytes = batch.words * 4 + batch.count * 8
    d_digests = dev.alloc(32 * n)
    d_roots = dev.alloc(32 * max(nslices, 1))
    scratch_bytes = (dev.lib.vkmr_hip_reduce_levels_scratch_bytes(cap) if a.levels_variant
                     else dev.lib.vkmr_hip_reduce_slices_scratch_bytes(cap, nslices))
    d_scratch = dev.alloc(scratch_bytes)
    d_top_scratch = dev.alloc(dev.lib.vkmr_hip_reduce_scratch_bytes(max(nslices, world, 2)) + 64)
    d_sub = dev.alloc(32)                       # this rank's sub-tree root: the 32 bytes that cross xGMI
    d_all = dev.alloc(32 * world)               # every rank's sub-tree root, in rank (= slice) order
    d_final = dev.alloc(32)
    # every slice of a multi-slice tree is reduced to capacity height (reference Reductions.cpp:471); a lone slice to its own
    slice_height = slice_log2 if (nslices * world > 1) else tree_height(n)

    ev = [(dev.new_event(), dev.new_event()) for _ in range(2 * a.steps * (nbatches + 2))]
    used = []
    final = np.zeros(8, dtype=np.uint32)
    host_sub = np.zeros(8, dtype=np.uint32)
    streams1 = (C.c_void_p * 1)(dev.stream)
    mine1 = (C.c_void_p * 1)(d_sub.ptr)
    all1 = (C.c_void_p * 1)(d_all.ptr)

    def timed(kind, on):
        if not on:
            return None
        e0, e1 = ev[len(used)]
        dev.record(e0)
        used.append((kind, e0, e1))
        return e1

    def step(on):
        # MAP: one launch per batch into its place in the slice(s)
        for b in range(nbatches):
            e1 = timed("map", on)
            d_data, words, d_meta = d_batches[b]
            dev.map_async(d_data, words, d_meta, bstr, d_digests, out_offset_digests=b * bstr)
            if e1:
                dev.record(e1)
        # REDUCE: this rank's leaves -> ONE sub-tree root (d_sub)
        e1 = timed("reduce", on)
        if a.levels_variant:
            for s in range(nslices):
                vk.check(dev.lib.vkmr_hip_reduce_levels_async(dev.index, dev.stream, d_digests.at(32 * s * cap), cap, slice_height,
                                                              d_scratch.ptr, (d_roots.at(32 * s) if nslices > 1 else d_sub.ptr)), "reduce_levels")
        elif nslices == 1:
            dev.reduce_async(d_digests, n, slice_height, d_scratch, d_sub)
        else:
            dev.reduce_slices_async(d_digests, nslices, cap, cap, slice_height, d_scratch, d_roots)
        if nslices > 1:   # several slices per GPU: their roots meet on the device first, so 32 bytes per GPU travel
            dev.reduce_async(d_roots, nslices, tree_height(nslices), d_top_scratch, d_sub)
        if e1:
            dev.record(e1)
        # GATHER + COMBINE: N sub-tree roots -> root
        if dist is None:
            vk.check(dev.lib.vkmr_hip_memcpy_d2h_async(dev.index, dev.stream, final.ctypes.data, d_sub.ptr, 32), "d2h")
            dev.sync()
        elif comm is not None:
            e1 = timed("gather", on)
            vk.check(dev.lib.vkmr_hip_gather_roots_async(comm, streams1, mine1, 1, all1), "vkmr_hip_gather_roots_async")   # ONE RCCL collective, 32 B per rank
            if e1:
                dev.record(e1)
            src = d_all
            if world > 1:
                dev.combine_async(d_all, world, d_top_scratch, d_final)
                src = d_final
            vk.check(dev.lib.vkmr_hip_memcpy_d2h_async(dev.index, dev.stream, final.ctypes.data, src.ptr, 32), "d2h")
            dev.sync()
        else:   # --rehearse-gloo
            from vk_merkle_roots_amd.distributed import gather_roots
            vk.check(dev.lib.vkmr_hip_memcpy_d2h_async(dev.index, dev.stream, host_sub.ctypes.data, d_sub.ptr, 32), "d2h")
            dev.sync()
            allr = gather_roots(host_sub.reshape(1, 8), dist, rank, world, device=None, equal_counts=True)
            if rank == 0:
                allr = np.ascontiguousarray(allr)
                vk.check(dev.lib.vkmr_hip_memcpy_h2d_async(dev.index, dev.stream, d_all.ptr, allr.ctypes.data, allr.nbytes), "h2d")
                dev.combine_async(d_all, world, d_top_scratch, d_final)
                vk.check(dev.lib.vkmr_hip_memcpy_d2h_async(dev.index, dev.stream, final.ctypes.data, d_final.ptr, 32), "d2h")
                dev.sync()

    def barrier():
        dev.sync()
        if dist is not None:
            import torch
            dist.barrier()
            if tdev is not None:
                torch.cuda.synchronize()
        dev.sync()

    # A multi-rank run that wedges (a rank lost, a collective that never completes) should say where, not sit until the
    # launcher's timeout: a watchdog thread ends the process with a message once a phase has taken implausibly long.
    phase = {"name": "warmup", "since": time.time()}
    if dist is not None:
        import threading

        def watchdog():
            while True:
                time.sleep(5)
                if phase["name"] == "done":
                    return
                if time.time() - phase["since"] > 240:
                    sys.stderr.write(f"[bench] rank {rank}: no progress for 240 s in phase '{phase['name']}' "
                                     f"(collective {collective}); last error of the C ABI: {vk._abi.what_error()!r}\n")
                    sys.stderr.flush()
                    os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()

    def enter(name):
        phase["name"], phase["since"] = name, time.time()

    # The GPU has idled through seconds of host-side generation and upload: it is in a low-power state, and the first tens of
    # milliseconds after it run on a clock that is still ramping (9.93 ms per step with --warmup 2 against 9.69 with 5 more
    # steps in front).  Half a second of untimed steps first, whatever W is; then the W warmup steps the contract names.
    # A FIXED number of them: every step of an N > 1 run holds a collective, so every rank must run the same count.
    for _ in range(48):
        step(False)
    for _ in range(a.warmup):
        step(False)
    enter("barrier before the timed steps")
    barrier()
    enter("timed steps")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    enter("after the timed steps")
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-kernel averages from the HIP events recorded on the launch stream
    def mean_ms(kind):
        v = [dev.elapsed_ms(e0, e1) for k, e0, e1 in used if k == kind]
        return float(np.mean(v)) if v else None
    map_launch_ms, red_step_ms, gather_ms = mean_ms("map"), mean_ms("reduce"), mean_ms("gather")

    if rank == 0:
        total_leaves = n * world
        ms_per_step = dt / a.steps * 1e3
        value = total_leaves * a.steps / dt
        root_hex = digest_hex(final)
        sub_roots = ([digest_hex(r) for r in dev.download(d_all, 32 * world).reshape(-1, 8)] if comm is not None or a.rehearse_gloo
                     else [digest_hex(dev.download(d_sub, 32))])
        # the timed path's own result against what the reference CPU path printed for these streams
        golden = golden_big_roots(a.leaves_log2, a.maxlen) if a.seed == 42 else None
        root_ok, subs_ok = check_against_golden(golden, a.seed, world, root_hex, sub_roots)
        # algorithmic bytes of one map launch: packed words + 8 B metadata read, 32 B digest written
        map_bytes = (batch.words * 4 + 8 * n + 32 * n) / nbatches
        achieved = map_bytes / (map_launch_ms * 1e-3) / 1e9
        # the reduction reads every digest once at tree level 0 (32 B per leaf); the levels above add 2/32 + 2/1024 + ... of that
        red_bytes = 32.0 * n * (1 + 2 / 32 + 2 / 1024)
        red_achieved = red_bytes / (red_step_ms * 1e-3) / 1e9
        sizes = batch.meta[:, 1].astype(np.int64)
        blocks = int(((sizes + 8) // 64 + 1).sum())
        kernel_info = dev.lib.vkmr_hip_kernel_info().decode()
        # HBM bytes per launch from the PMC passes of the SAME launch shape, the SAME kernel instantiation and the SAME
        # build of the kernel sources (profiles/pmc_latest.json, tools/pmc_profile.sh + tools/pmc_to_json.py on the
        # GPU box); null with the reason otherwise
        from vk_merkle_roots_amd import provenance
        pmc = provenance.load_pmc()
        traffic, traffic_src = provenance.traffic_from_pmc(pmc, kernel_info, "map_kernel_hbm_bytes_per_launch", strings_per_map_launch=bstr, maxlen=a.maxlen)
        red_traffic, red_traffic_src = provenance.traffic_from_pmc(pmc, kernel_info, "reduce_hbm_bytes_per_step", strings_per_map_launch=bstr, maxlen=a.maxlen,
                                                                   slice_log2=slice_log2)
        out = {
            "metric": "leaf hashes/sec (SHA-256d) + Merkle-root wall time, 2^26 leaves, 1/2/4/8 GPU",
            "value": value, "unit": "leaf hashes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload_label(a.seed, a.leaves_log2, a.maxlen, world),
                       "leaves_per_gpu": n, "leaves_total": total_leaves, "slices_per_gpu": nslices, "slice_capacity": cap,
                       "map_launches_per_step": nbatches, "input_bytes_per_gpu": int(input_bytes),
                       "parallelism": f"one process per GPU, slices sharded over {world} GPU(s), no data-path collective but the root gather",
                       "collective": collective, "ranks": world,
                       "gather": gather_label(dist is not None, comm is not None),
                       "bytes_gathered_per_step": (32 * world if dist is not None else 0),
                       "kernels": kernel_info, "reduce_variant": "levels" if a.levels_variant else "wave",
                       "rccl": dev.lib.vkmr_hip_comm_info().decode()},
            "root": root_hex,
            "sub_roots": sub_roots,
            "root_matches_golden": root_ok,
            "sub_roots_match_golden": subs_ok,
            "golden": "tests/golden/big_roots.json (reference CPU path on the same rndm streams)" if golden else None,
            "merkle_root_wall_ms": ms_per_step,
            "roofline": {"bound": "hbm", "kernel": "map_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "launch_ms": map_launch_ms, "algorithmic_bytes_per_launch": map_bytes},
            "roofline_reduce": {"bound": "hbm", "kernel": "reduce_pass_kernel + reduce_collapse_kernel + reduce_tail_kernel (one slice reduction)",
                                "achieved": red_achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": red_achieved / HBM_PEAK_GBPS,
                                "traffic": red_traffic, "traffic_source": red_traffic_src, "step_ms": red_step_ms, "algorithmic_bytes_per_step": red_bytes},
            "valu_roofline": None,   # filled below (needs the clock leg, which runs after the resident buffers are freed)
            "gather_ms": gather_ms,
            "setup": {"generate_s": t_gen, "h2d_pageable_s": t_h2d},
        }
        if world == 1 and not a.no_pipeline and not a.levels_variant and nbatches == 1 and nslices == 1 and a.leaves_log2 > 23:
            # the same step in the reference's shapes -- map launches of 2^23 strings (its batch, Batches.h:131-134), each its
            # OWN packed batch (own data words: the launch picks its fetch mode from them, as a real 2^23-string batch would),
            # slices of 2^23 digests (its slice, SHA-256vk.cpp:23) reduced by one batched call, roots combined on the device
            sub, ns = 1 << 23, n >> 23
            subs = []
            for b in range(ns):
                sb = batch.slice(b * sub, (b + 1) * sub)
                subs.append((dev.upload(sb.data), sb.words, dev.upload(sb.meta)))
            d_r8 = dev.alloc(32 * ns)
            d_s8 = dev.alloc(dev.lib.vkmr_hip_reduce_slices_scratch_bytes(sub, ns))

            def ref_step():
                for b, (sd, sw, sm) in enumerate(subs):
                    dev.map_async(sd, sw, sm, sub, d_digests, out_offset_digests=b * sub)
                dev.reduce_slices_async(d_digests, ns, sub, sub, 23, d_s8, d_r8)
                dev.combine_async(d_r8, ns, d_top_scratch, d_final)
                vk.check(dev.lib.vkmr_hip_memcpy_d2h_async(dev.index, dev.stream, final.ctypes.data, d_final.ptr, 32), "d2h")
                dev.sync()
            ref_step()
            ref_mode = dev.lib.vkmr_hip_kernel_info().decode().split(" reduce=")[0]
            t0r = time.perf_counter()
            for _ in range(5):
                ref_step()
            dtr = (time.perf_counter() - t0r) / 5
            out["reference_shapes"] = {"map_launches_per_step": ns, "slices": ns, "ms_per_step": dtr * 1e3, "leaf_hashes_per_s": n / dtr,
                                       "root_matches": digest_hex(final) == root_hex, "map_mode": ref_mode,
                                       "what": "8 packed batches of 2^23 strings (own buffers), slices of 2^23 digests (the reference's shapes), resident input"}
            for sd, _, sm in subs:
                sd.free()
                sm.free()
            d_r8.free()
            d_s8.free()
            dev.map_async(d_batches[0][0], d_batches[0][1], d_batches[0][2], bstr, d_digests)   # kernel_info reports the main shape again
            dev.sync()
        if world == 1 and not a.no_pipeline and not a.levels_variant and nbatches == 1 and nslices == 1:
            ts = two_stream_rate(dev, vk, d_batches[0], bstr, n, slice_height)
            out["two_stream_overlap"] = {"leaf_hashes_per_s": ts["leaf_hashes_per_s"], "ms_per_step": ts["ms_per_step"], "what": ts["what"],
                                         "roots_match": all(digest_hex(r) == root_hex for r in ts["roots"])}
        if world == 1:
            for b_ in d_batches:    # the resident workload is no longer needed: give the HBM back before the secondary legs
                b_[0].free()
                b_[2].free()
            d_digests.free()
        if world == 1 and not a.no_pipeline and not a.levels_variant:
            pl_bstr = min(n, 1 << 23)   # the stream processor's shape: batches small enough for copies to hide behind kernels
            pl = pipeline_rate(vk, batch, pl_bstr, slice_log2 if nslices > 1 else min(slice_log2, 23), device=local_rank)
            out["pipeline_pcie_inclusive"] = {"leaf_hashes_per_s": pl["leaf_hashes_per_s"], "ms": pl["ms"], "h2d_GBps": pl["h2d_GBps"],
                                              "h2d_bytes_per_string_of_metadata": pl["h2d_bytes_per_string_of_metadata"],
                                              "root_matches": pl["root_hex"] == root_hex, "runs_ms": pl["runs_ms"],
                                              "what": "libvkmr_pipeline.so: strings staged in the C++ stream processor's pinned batches, then its own "
                                                      "schedule -- per batch H2D on the copy stream and the map kernel behind it, slices of 2^23 to "
                                                      "Reductions as they fill, combine on the device (Mappings::Map / Reductions of `vkmr hip:0`)"}
        if world == 1 and not a.no_pipeline and not a.levels_variant:
            try:
                ff = from_file_rate(a.seed, n, a.maxlen, local_rank)
            except Exception as e:   # no room for the stream in the temporary directory, a time-out ...: the leg is reported as failed, the line is still printed
                ff = {"error": f"{type(e).__name__}: {e}"[:300]}
            if "root_hex" in ff:
                ff["root_matches"] = ff.pop("root_hex") == root_hex and ff["items"] == n
            out["from_file_on_stdin"] = ff
        if world == 1 and not a.no_long_strings:
            out["long_strings"] = long_strings_rate(dev, vk, a.seed)
        if world == 1 and not a.no_config5 and not a.no_long_strings and a.leaves_log2 >= 24:
            try:
                out["config5_full"] = config5_full(dev, vk, a.seed)
            except Exception as e:   # out of host or device memory on an unusual box: the leg is reported as failed, the line is still printed
                out["config5_full"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # ---- the bound the path sits under: VALU issue at the clock each kernel holds -------------------------------
        def valu_roofline_block(clk):
            counts = static_counts()
            vr = {"bound": "VALU issue: a gfx950 SIMD issues at most two VALU instructions per 4-cycle turn -- one of any kind plus one simple one "
                           "(add/sub, and/or/xor, shift right, v_bitop3 on VGPRs); floor = max(N / 2, N_complex) turns per hash (DESIGN.md 3)",
                  "static_counts": counts, "clock": clk,
                  "map_ms_per_step": map_launch_ms * nbatches, "reduce_ms_per_step": red_step_ms}
            if counts and counts.get("node"):
                mode = provenance.map_symbol_of(kernel_info)
                key = mode[len("map_kernel<"):-1].replace("false", "0").replace("true", "1") if mode else None   # as the mangled names spell it
                mc = counts["map"].get(key) if key else None
                ghz = lambda k: (clk or {}).get(k, {}).get("GHz_median") if clk and "error" not in clk else None   # noqa: E731
                node = counts["node"]
                vr["reduce"] = issue_block(n - 1, node["valu"], max(node["valu"] / 2.0, node["complex"]), red_step_ms, ghz("reduce_pass_kernel"))
                if mc:
                    bpl = blocks / n
                    valu = bpl * mc["block"]["valu"] + mc["digest"]["valu"]
                    cplx = bpl * mc["block"]["complex"] + mc["digest"]["complex"]
                    vr["map"] = issue_block(n, valu, max(valu / 2.0, cplx), map_launch_ms * nbatches, ghz("map_kernel"))
                    vr["map"]["blocks_per_leaf"] = bpl
            return vr

        def cpu_baseline_block():
            cb = cpu_baseline(a.seed, a.maxlen, a.leaves_log2 if a.cpu_sample_log2 is None else min(a.cpu_sample_log2, a.leaves_log2), a.leaves_log2)
            if world > 1:
                cb["sample"] = f"rank 0's stream only -- 1/{world} of the workload: " + cb["sample"]
            want = sub_roots[0] if world > 1 else root_hex      # rank 0's sub-tree root is the root of exactly this stream
            if cb.get("root") and cb["leaves"] == n:
                cb["root_matches_gpu"] = cb["root"] == want
            return cb

        if world == 1:
            out["valu_roofline"] = valu_roofline_block(None if a.no_clock_leg else clock_leg(a.seed, a.maxlen, a.leaves_log2))
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_block()
    # Every rank leaves the data path together BEFORE rank 0 runs its one-process probe: the other ranks must not sit in
    # a barrier under the watchdog while rank 0 spends up to two minutes in child processes, and must not tear their
    # communicators down while it still uses the GPUs.
    enter("teardown")
    if comm is not None:
        dev.lib.vkmr_hip_comm_destroy(comm)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1:
            # An N > 1 line carries what the N = 1 line carries: the in-kernel clock (rank 0's GPU, the same
            # per-GPU workload) and the reference's CPU path timed on rank 0's stream -- after every rank has left the data
            # path and given its GPU back, so that nobody waits under a watchdog while this runs.
            enter("clock leg")
            for buf in [b_[0] for b_ in d_batches] + [b_[2] for b_ in d_batches] + [d_digests]:
                buf.free()
            out["valu_roofline"] = valu_roofline_block(None if (a.no_clock_leg or collective != "nccl") else clock_leg(a.seed, a.maxlen, a.leaves_log2))
            if not a.no_cpu_baseline:
                enter("cpu baseline")
                out["cpu_baseline"] = cpu_baseline_block()
        if world > 1 and collective == "nccl":
            enter("hip:all probe")
            out["hip_all_one_process_check"] = hip_all_check(timeout_s=60)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    enter("done")


if __name__ == "__main__":
    main()
