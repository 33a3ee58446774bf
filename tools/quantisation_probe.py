#!/usr/bin/env python3
"""What does the quantisation of a tile's 16 groups over its 8 wavefronts cost the staged map kernel?  
The kernel hands groups of 64 length-sorted strings to wavefronts longest first; a group costs blocks + 1 compressions.  With
`rndm 42 * 127` a tile's groups cost 4, 3 x8, 2 x7 = 42 units: 5.25 per wavefront, but nobody finishes before 6.  Streams whose
lengths make the groups fit exactly (every string 64 B: 16 x 3 = 8 x 6; 80 B and 40 B alternating: 8 x (3 + 2); all within the 17 words per string a tile stages) have no such
loss.  Compared: compressions per microsecond.  If the exact fits are ~12 % better per compression the loss is real and
recoverable; if they are the same, the board gives the time back as clock.   GPU box:  python3 tools/quantisation_probe.py"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vk_merkle_roots_amd as vk  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2", type=int, default=24)
a = ap.parse_args()
dev = vk.HipDevice(0)
n = 1 << a.log2
rng = np.random.default_rng(5)


def blocks(size):
    return (size + 8) // 64 + 1


def batch_of(sizes):
    sizes = np.asarray(sizes, dtype=np.uint32)
    words = (sizes.astype(np.uint64) + 3) // 4
    start = np.concatenate([[0], np.cumsum(words)[:-1]]).astype(np.uint32)
    total = int(words.sum())
    data = rng.integers(0, 2**32, size=total, dtype=np.uint32)
    meta = np.stack([start, sizes], axis=1).astype(np.uint32)
    return vk.PackedBatch(data, meta, total, int(sizes.astype(np.uint64).sum()))


def tile_makespan(sizes, tile=1024, waves=8):
    """Mean over the first tiles of (units per wavefront on average, units of the busiest wavefront) under longest-first hand-out."""
    avg, mx = [], []
    for t in range(0, min(len(sizes), 64 * tile), tile):
        s = np.sort(sizes[t:t + tile])[::-1]
        cost = [int(blocks(int(s[g])) + 1) for g in range(0, len(s), 64)]     # sorted: the group's first string is its longest
        busy = [0] * waves
        for c in cost:
            busy[busy.index(min(busy))] += c
        avg.append(sum(cost) / waves)
        mx.append(max(busy))
    return float(np.mean(avg)), float(np.mean(mx))


b0 = vk.rndm_packed(42, n, 127)
i = np.arange(n)
cases = {
    "rndm 42 127 (the headline stream)": b0,
    "every string 64 B (16 x 3)": batch_of(np.full(n, 64)),
    "every string 40 B (16 x 2)": batch_of(np.full(n, 40)),
    "80 B / 40 B alternating (8 x (3 + 2))": batch_of(np.where(i % 2 == 0, 80, 40)),
    "80 B / 40 B, 9 : 7 per tile (9 x 3 + 7 x 2 = 41)": batch_of(np.where(i % 16 < 9, 80, 40)),
    "80 B / 40 B, 10 : 6 per tile (10 x 3 + 6 x 2 = 42)": batch_of(np.where(i % 16 < 10, 80, 40)),
    "40 B, and 64 strings of 120 B per tile (4 + 15 x 2 = 34: the 16th group makes one wavefront do 6)": batch_of(np.where(i % 1024 < 64, 120, 40)),
    "40 B, and 128 strings of 120 B per tile (2 x 4 + 14 x 2 = 36)": batch_of(np.where(i % 1024 < 128, 120, 40)),
    "rndm lengths, same sizes sorted inside each tile": None,
}
sz = b0.meta[:, 1].copy()
cases["rndm lengths, same sizes sorted inside each tile"] = batch_of(np.sort(sz.reshape(-1, 1024), axis=1)[:, ::-1].reshape(-1))
e0, e1 = dev.new_event(), dev.new_event()
print(f"{n} strings per launch, staged map kernel; median of 9 launches after 6")
for name, b in cases.items():
    sizes = b.meta[:, 1].astype(np.int64)
    units = int((blocks(sizes) + 1).sum())
    avg, mx = tile_makespan(sizes)
    d_data, d_meta, d_out = dev.upload(b.data), dev.upload(b.meta), dev.alloc(32 * n)
    for _ in range(6):
        dev.map_async(d_data, b.words, d_meta, n, d_out)
    dev.sync()
    t = []
    for _ in range(9):
        dev.record(e0); dev.map_async(d_data, b.words, d_meta, n, d_out); dev.record(e1); dev.sync()
        t.append(dev.elapsed_ms(e0, e1))
    ms = float(np.median(t))
    print(f"{name:100s} {ms:7.3f} ms  {units / n:5.2f} compressions/string  {units / ms / 1e3:8.1f} compressions/us   "
          f"tile: {avg:5.2f} units per wavefront, busiest {mx:5.2f} ({avg / mx:.3f})  avg {sizes.mean():5.1f} B")
    d_data.free(); d_meta.free(); d_out.free()
