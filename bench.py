#!/usr/bin/env python3
"""bench.py -- leaf hashes/s and Merkle-root wall time on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--leaves-log2 26] [--maxlen 127]

One "step" = one pass of the hot path (map -> reduce -> combine) over the whole
synthetic workload, with the packed batches already resident in HBM.  The workload
is BASELINE.json's configs[2]: `rndm <seed> 2^26 127` (the restated glibc rand()
generator in csrc/host/rndm_stream.cpp produces the identical strings straight into
packed batches).  With N > 1 (launched by torch.distributed.run, one rank per GPU)
every rank holds its own 2^26 leaves (weak scaling: slices sharded across GPUs, no
data-path collective), reduces them to slice roots, and the roots are gathered once
over RCCL and combined on rank 0.

Prints ONE JSON line on rank 0 (see the driver contract): `value` is whole-job leaf
hashes/s; `roofline` prices the dominant kernel against HBM (the spec'd bound) and
carries the int32-VALU bound the path actually sits under; `cpu_baseline` is the
reference's own CPU-serial path (oracle/_ref, built from the reference sources)
timed on a bounded prefix of the same stream on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0            # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12   # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T int32 lane-ops/s


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--leaves-log2", type=int, default=26, help="leaves per GPU (default 2^26 = configs[2])")
    p.add_argument("--maxlen", type=int, default=127, help="rndm max string length argument")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--slice-log2", type=int, default=23, help="digests per slice (reference: 2^23 = 256 MiB)")
    p.add_argument("--batch-log2", type=int, default=23, help="strings per map launch (reference: <= 2^23 per batch)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-log2", type=int, default=22, help="prefix of the stream given to the CPU baseline")
    p.add_argument("--levels-variant", action="store_true", help="use the one-level-per-launch reduction")
    p.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for rehearsal)")
    p.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with one rank (plumbing check)")
    p.add_argument("--no-pipeline", action="store_true", help="skip the PCIe-inclusive (pinned host -> root) measurement")
    return p.parse_args()


def cpu_baseline(seed, maxlen, sample_log2):
    """The reference CPU-serial path (oracle/_ref/vkmr_cpu_ref: the reference's own
    SHA-256plus/Inputs/StopWatch sources, g++ -O2) on the first 2^sample_log2 strings
    of the same rndm stream, timed by its own stopwatch line (reference Vkmr.cpp:55).
    Falls back to the C restatement ("port") when the reference build is absent."""
    n = 1 << sample_log2
    rndm = os.path.join(ROOT, "vk_merkle_roots_amd", "bin", "rndm")
    ref = os.path.join(ROOT, "oracle", "_ref", "vkmr_cpu_ref")
    sample = f"first 2^{sample_log2} strings of rndm {seed} * {maxlen} via stdin"
    if os.path.exists(ref) and os.path.exists(rndm):
        gen = subprocess.Popen([rndm, str(seed), str(n), str(maxlen)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        out = subprocess.run([ref], stdin=gen.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        gen.wait()
        line = [l for l in out.splitlines() if "computed root" in l]
        if line:
            ms = float(line[0].rsplit(" in ", 1)[1])
            return {"value": n / (ms / 1e3), "unit": "leaf hashes/s", "cores": 1, "kind": "reference",
                    "sample": sample + " (stdin parse + hash + tree, program's own stopwatch; g++ -O2)",
                    "host_cores": os.cpu_count(), "seconds": ms / 1e3}
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
            "sample": sample.replace("via stdin", "packed") + " (hash + tree only)", "host_cores": os.cpu_count(), "seconds": dt}


def pipeline_rate(dev, vk, batch, nbatches, bstr, cap, nslices, slice_height, tree_height, steps=3):
    """Pipeline-level rate (SURVEY.md 8d ii): packed batches in PINNED host memory -> root, H2D copies
    on a copy stream overlapped with the map kernels on the compute stream (two HBM landing zones),
    then the batched reduction and the combine.  PCIe-inclusive; reported beside `value`, never as it."""
    n = batch.count
    subs = [batch.slice(b * bstr, (b + 1) * bstr) for b in range(nbatches)]
    pinned = []
    for sub in subs:
        pd, pm = C.c_void_p(), C.c_void_p()
        vk.check(dev.lib.vkmr_hip_host_alloc(max(sub.words * 4, 4), C.byref(pd)), "host_alloc")
        vk.check(dev.lib.vkmr_hip_host_alloc(sub.count * 8, C.byref(pm)), "host_alloc")
        C.memmove(pd.value, sub.data.ctypes.data, sub.words * 4)
        C.memmove(pm.value, sub.meta.ctypes.data, sub.count * 8)
        pinned.append((pd.value, pm.value, sub.words))
    zmax = max(w for _, _, w in pinned)
    zones = [(dev.alloc(zmax * 4), dev.alloc(bstr * 8)) for _ in range(2)]
    d_digests = dev.alloc(32 * n)
    d_roots = dev.alloc(32 * nslices)
    d_scratch = dev.alloc(dev.lib.vkmr_hip_reduce_slices_scratch_bytes(cap, nslices))
    d_top = dev.alloc(dev.lib.vkmr_hip_reduce_scratch_bytes(max(nslices, 2)) + 64)
    d_final = dev.alloc(32)
    copy_stream = dev.new_stream()
    ev_copied = [dev.new_event() for _ in range(2)]
    ev_mapped = [dev.new_event() for _ in range(2)]
    final = np.zeros(8, dtype=np.uint32)

    def run():
        for b, (pd, pm, words) in enumerate(pinned):
            z = b & 1
            zd, zm = zones[z]
            if b >= 2:   # the landing zone is free once the map that read it has finished
                vk.check(dev.lib.vkmr_hip_stream_wait_event(dev.index, copy_stream, ev_mapped[z]), "wait")
            vk.check(dev.lib.vkmr_hip_memcpy_h2d_async(dev.index, copy_stream, zd.ptr, pd, words * 4), "h2d")
            vk.check(dev.lib.vkmr_hip_memcpy_h2d_async(dev.index, copy_stream, zm.ptr, pm, bstr * 8), "h2d")
            dev.record(ev_copied[z], copy_stream)
            vk.check(dev.lib.vkmr_hip_stream_wait_event(dev.index, dev.stream, ev_copied[z]), "wait")
            dev.map_async(zd, words, zm, bstr, d_digests, out_offset_digests=b * bstr)
            dev.record(ev_mapped[z])
        dev.reduce_slices_async(d_digests, nslices, cap, cap, slice_height, d_scratch, d_roots)
        src = d_roots
        if nslices > 1:
            dev.reduce_async(d_roots, nslices, tree_height(nslices), d_top, d_final)
            src = d_final
        vk.check(dev.lib.vkmr_hip_memcpy_d2h_async(dev.index, dev.stream, final.ctypes.data, src.ptr, 32), "d2h")
        dev.sync()
        dev.sync(copy_stream)

    run()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    dt = (time.perf_counter() - t0) / steps
    for pd, pm, _ in pinned:
        dev.lib.vkmr_hip_host_free(pd)
        dev.lib.vkmr_hip_host_free(pm)
    for b in (d_digests, d_roots, d_scratch, d_top, d_final, zones[0][0], zones[0][1], zones[1][0], zones[1][1]):
        b.free()
    return {"leaf_hashes_per_s": n / dt, "ms": dt * 1e3, "h2d_GBps": (batch.words * 4 + n * 8) / dt / 1e9, "root": final.copy()}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    tdev = None
    collective = None
    if world > 1 or a.force_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        import datetime
        collective = a.dist_backend
        if a.dist_backend == "nccl":
            try:
                torch.cuda.set_device(local_rank)
                tdev = torch.device("cuda", local_rank)
                dist.init_process_group(backend="nccl", device_id=tdev, rank=rank, world_size=world,
                                        timeout=datetime.timedelta(seconds=300))
                probe = torch.ones(1, device=tdev)
                dist.all_reduce(probe)            # first RCCL collective: fail here, not inside the timed region
                torch.cuda.synchronize()
            except Exception as e:                # RCCL unusable on this node: the 32-byte gather goes over gloo, and says so
                sys.stderr.write(f"[bench] nccl/RCCL init failed on rank {rank}: {e!r}; falling back to gloo for the root gather\n")
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                tdev = None
                collective = "gloo (RCCL init failed)"
                dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend=a.dist_backend, rank=rank, world_size=world)
            local_rank = local_rank % max(1, torch.cuda.device_count())
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import vk_merkle_roots_amd as vk
    from vk_merkle_roots_amd.engine import digest_hex, tree_height

    dev = vk.HipDevice(local_rank)
    n = 1 << a.leaves_log2
    cap = 1 << min(a.slice_log2, a.leaves_log2)
    bstr = 1 << min(a.batch_log2, a.leaves_log2)
    nslices = n // cap
    nbatches = n // bstr

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
    d_top_scratch = dev.alloc(dev.lib.vkmr_hip_reduce_scratch_bytes(max(nslices * world, 2)) + 64)
    d_all_roots = dev.alloc(32 * nslices * world)
    d_final = dev.alloc(32)
    slice_height = min(a.slice_log2, a.leaves_log2) if (nslices * world > 1) else tree_height(n)

    ev = [(dev.new_event(), dev.new_event()) for _ in range(2 * a.steps * (nbatches + 1))]
    used = []
    host_roots = np.zeros((nslices, 8), dtype=np.uint32)
    final = np.zeros(8, dtype=np.uint32)
    if dist is not None:
        from vk_merkle_roots_amd.distributed import gather_roots

    def step(timed):
        # MAP: one launch per batch into its place in the slice(s)
        for b in range(nbatches):
            if timed:
                e0, e1 = ev[len(used)]
                dev.record(e0)
            d_data, words, d_meta = d_batches[b]
            dev.map_async(d_data, words, d_meta, bstr, d_digests, out_offset_digests=b * bstr)
            if timed:
                dev.record(e1)
                used.append(("map", e0, e1))
        # REDUCE: every slice to its root
        if timed:
            e0, e1 = ev[len(used)]
            dev.record(e0)
        if a.levels_variant:
            for s in range(nslices):
                vk.check(dev.lib.vkmr_hip_reduce_levels_async(dev.index, dev.stream, d_digests.at(32 * s * cap), cap, slice_height,
                                                              d_scratch.ptr, d_roots.at(32 * s)), "reduce_levels")
        else:
            dev.reduce_slices_async(d_digests, nslices, cap, cap, slice_height, d_scratch, d_roots)
        if timed:
            dev.record(e1)
            used.append(("reduce", e0, e1))
        # COMBINE: slice roots -> root (on device for one GPU; one RCCL gather for several)
        if dist is None:
            if nslices > 1:
                dev.reduce_async(d_roots, nslices, tree_height(nslices), d_top_scratch, d_final)
                src = d_final
            else:
                src = d_roots
            vk.check(dev.lib.vkmr_hip_memcpy_d2h_async(dev.index, dev.stream, final.ctypes.data, src.ptr, 32), "d2h")
            dev.sync()
        else:
            vk.check(dev.lib.vkmr_hip_memcpy_d2h_async(dev.index, dev.stream, host_roots.ctypes.data, d_roots.ptr, 32 * nslices), "d2h")
            dev.sync()
            allr = gather_roots(host_roots, dist, rank, world, device=tdev, equal_counts=True)   # ONE RCCL gather, 32 B per slice
            if rank == 0:
                allr = np.ascontiguousarray(allr)
                vk.check(dev.lib.vkmr_hip_memcpy_h2d_async(dev.index, dev.stream, d_all_roots.ptr, allr.ctypes.data, allr.nbytes), "h2d")
                dev.reduce_async(d_all_roots, allr.shape[0], tree_height(allr.shape[0]), d_top_scratch, d_final)
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

    for _ in range(a.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-kernel averages from the HIP events recorded on the launch stream
    map_ms = [dev.elapsed_ms(e0, e1) for k, e0, e1 in used if k == "map"]
    red_ms = [dev.elapsed_ms(e0, e1) for k, e0, e1 in used if k == "reduce"]
    map_launch_ms = float(np.mean(map_ms))
    red_step_ms = float(np.mean(red_ms))

    if rank == 0:
        total_leaves = n * world
        ms_per_step = dt / a.steps * 1e3
        value = total_leaves * a.steps / dt
        # algorithmic bytes of one map launch: packed words + 8 B metadata read, 32 B digest written
        map_bytes = (batch.words * 4 + 8 * n + 32 * n) / nbatches
        achieved = map_bytes / (map_launch_ms * 1e-3) / 1e9
        # int32 VALU work in full-rate issue slots (static counts of the shipped code, tools/isa_count.py):
        # data block 2202 (compression) + ~330 (fetch, byte swap, padding masks), digest hash 2093, tree node 5685
        sizes = batch.meta[:, 1].astype(np.int64)
        blocks = int(((sizes + 8) // 64 + 1).sum())
        map_ops = blocks * 2532 + n * 2093
        red_ops = (n - nslices) * 5685
        # HBM bytes per map launch from the PMC passes of the SAME launch shape (profiles/pmc_latest.json,
        # written by tools/pmc_profile.sh + tools/pmc_to_json.py on the GPU box); null when none matches
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("strings_per_map_launch") == bstr and rec.get("maxlen") == a.maxlen and not os.environ.get("VKMR_MAP_VARIANT"):
                    traffic = rec.get("map_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "leaf hashes/sec (SHA-256d) + Merkle-root wall time, 2^26 leaves, 1/2/4/8 GPU",
            "value": value, "unit": "leaf hashes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"rndm {a.seed}+rank 2^{a.leaves_log2} {a.maxlen} per GPU (BASELINE configs[2])",
                       "leaves_per_gpu": n, "slices_per_gpu": nslices, "slice_capacity": cap, "map_launches_per_step": nbatches,
                       "input_bytes_per_gpu": int(input_bytes), "parallelism": f"slices sharded over {world} GPU(s)", "collective": (collective if dist is not None else None),
                       "kernels": dev.lib.vkmr_hip_kernel_info().decode(), "reduce_variant": "levels" if a.levels_variant else "wave"},
            "root": digest_hex(final),
            "merkle_root_wall_ms": ms_per_step,
            "roofline": {"bound": "hbm", "kernel": "map_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "launch_ms": map_launch_ms, "algorithmic_bytes_per_launch": map_bytes},
            "valu_roofline": {"bound": "int32-valu issue slots (v_alignbit/v_add3/v_perm count 2)", "sustained_tops_measured": 64.0, "map_ms_per_step": map_launch_ms * nbatches, "reduce_ms_per_step": red_step_ms,
                              "map_achieved_tops": map_ops / (map_launch_ms * nbatches * 1e-3) / 1e12,
                              "reduce_achieved_tops": red_ops / (red_step_ms * 1e-3) / 1e12,
                              "peak_tops": VALU_PEAK_TOPS,
                              "map_frac": map_ops / (map_launch_ms * nbatches * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                              "reduce_frac": red_ops / (red_step_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS},
            "setup": {"generate_s": t_gen, "h2d_pageable_s": t_h2d},
        }
        if world == 1 and not a.no_pipeline and not a.levels_variant:
            pl = pipeline_rate(dev, vk, batch, nbatches, bstr, cap, nslices, slice_height, tree_height)
            out["pipeline_pcie_inclusive"] = {"leaf_hashes_per_s": pl["leaf_hashes_per_s"], "ms": pl["ms"], "h2d_GBps": pl["h2d_GBps"],
                                              "root_matches": digest_hex(pl["root"]) == digest_hex(final),
                                              "what": "pinned host batches -> async H2D overlapped with map -> reduce -> root"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.seed, a.maxlen, min(a.cpu_sample_log2, a.leaves_log2))
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
