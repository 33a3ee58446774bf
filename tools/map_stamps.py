#!/usr/bin/env python3
"""Diagnostic: where a map workgroup spends its shader clocks (needs build/ab/libstamps.so,
built with -DVKMR_MAP_STAMPS; stamps go to a per-workgroup debug slot, no atomics).
Read the SHARES; the stamped build is not the product."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VKMR_HIP_LIB"] = os.path.join(ROOT, "build", "ab", "libstamps.so")
import vk_merkle_roots_amd as vk  # noqa: E402

dev = vk.HipDevice(0)
log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 23
maxlen = int(sys.argv[2]) if len(sys.argv) > 2 else 127
b = vk.rndm_packed(42, 1 << log2, maxlen)
d_data, d_meta, d_out = dev.upload(b.data), dev.upload(b.meta), dev.alloc(32 * b.count)
L = C.CDLL(os.environ["VKMR_HIP_LIB"])
N = 32768
buf = np.zeros(N * 8, dtype=np.uint64)
e0, e1 = dev.new_event(), dev.new_event()
for it in range(3):
    dev.record(e0)
    dev.map_async(d_data, b.words, d_meta, b.count, d_out)
    dev.record(e1)
    dev.sync()
    L.vkmr_hip_debug_stamps(C.c_void_p(buf.ctypes.data), N * 8)
    s = buf.reshape(N, 8)
    s = s[s[:, 3] > 0].astype(np.float64)
    sort, stage, hsh = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    span_clk = s[:, 3].max() - s[:, 0].min()
    span_rt = (s[:, 4].max() - s[:, 4].min()) / 100e6
    print(f"iter {it}: kernel {dev.elapsed_ms(e0, e1):.3f} ms, {len(s)} workgroups; wavefront 0 medians: sort {np.median(sort):.0f} "
          f"stage {np.median(stage):.0f} hash {np.median(hsh):.0f} clocks; s_memtime span {span_clk:.3g} ticks over {span_rt * 1e3:.3f} ms "
          f"=> {span_clk / max(span_rt, 1e-9) / 1e9:.2f} GHz")
