#!/usr/bin/env python3
"""One slice of 2^k digests reduced with and without Merkle proofs written in the pass (vkmr_hip_reduce_proofs_async), and the
recomputing form (vkmr_hip_proof_async) beside them.  GPU box.   python3 tools/proof_timing.py [--log2 26] [--proofs 8]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vk_merkle_roots_amd as vk  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2", type=int, default=26)
ap.add_argument("--proofs", type=int, default=8)
ap.add_argument("--runs", type=int, default=12)
a = ap.parse_args()
dev = vk.HipDevice(0)
n = 1 << a.log2
rng = np.random.default_rng(7)
d_in = dev.alloc(32 * n)
chunk = 1 << 22
for at in range(0, n, chunk):   # random digests, uploaded in pieces
    part = rng.integers(0, 2**32, size=(min(chunk, n - at), 8), dtype=np.uint32)
    vk.check(dev.lib.vkmr_hip_memcpy_h2d_async(dev.index, dev.stream, d_in.at(32 * at), part.ctypes.data, part.nbytes), "h2d")
    dev.sync()
height = a.log2
d_scr, d_root, d_root2 = dev.reduce_scratch(n), dev.alloc(32), dev.alloc(32)
idx = np.ascontiguousarray(rng.integers(0, n, size=a.proofs), dtype=np.uint64)
d_sib = dev.alloc(32 * height * a.proofs)
runs = max(a.runs, 8)
ev = {k: [(dev.new_event(), dev.new_event()) for _ in range(runs)] for k in ("plain", "proofs", "old")}


def plain():
    dev.reduce_async(d_in, n, height, d_scr, d_root)


def with_proofs():
    vk.check(dev.lib.vkmr_hip_reduce_proofs_async(dev.index, dev.stream, d_in.ptr, n, height, d_scr.ptr, d_root2.ptr, idx.ctypes.data, a.proofs, d_sib.ptr), "reduce_proofs")


def recompute():
    vk.check(dev.lib.vkmr_hip_proof_async(dev.index, dev.stream, d_in.ptr, n, height, int(idx[0]), d_scr.ptr, d_sib.ptr, None), "proof")


# warm up until the clocks have settled, then the three forms in turn, run after run: a form timed alone, before or after the
# others, is compared across a clock that drifts by several per cent (the same reduction: 4.46 then 4.22 ms)
for _ in range(40):
    plain(); with_proofs()
dev.sync()
for r in range(runs):
    for name, fn in (("plain", plain), ("proofs", with_proofs), ("old", recompute)):
        e0, e1 = ev[name][r]
        dev.record(e0); fn(); dev.record(e1)
dev.sync()
med = {k: float(np.median([dev.elapsed_ms(e0, e1) for e0, e1 in v])) for k, v in ev.items()}
plain_ms, proofs, old = med["plain"], med["proofs"], med["old"]
same = (dev.download(d_root, 32) == dev.download(d_root2, 32)).all()
print(f"2^{a.log2} digests, medians of {runs} interleaved runs: reduce {plain_ms:.3f} ms; reduce writing {a.proofs} proofs in the pass {proofs:.3f} ms "
      f"({(proofs / plain_ms - 1) * 100:+.2f} %); one proof by re-reducing its sibling sub-trees (vkmr_hip_proof_async) {old:.3f} ms; same root: {bool(same)}")
