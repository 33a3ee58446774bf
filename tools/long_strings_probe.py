"""Map kernel on rndm 42 2^21 4096 (one batch), a few launches; prints ms per launch.  For A/B of experiment knobs
(VKMR_MAP_DYNLDS, VKMR_MAP_TILE, VKMR_MAP_VARIANT) and for rocprofv3 --pmc passes.  GPU box only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import vk_merkle_roots_amd as vk  # noqa: E402

dev = vk.HipDevice(0)
log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 21
maxlen = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
b = vk.rndm_packed(42, 1 << log2, maxlen)
d_data, d_meta, d_out = dev.upload(b.data), dev.upload(b.meta), dev.alloc(32 * b.count)
ev = [(dev.new_event(), dev.new_event()) for _ in range(10)]
# warm up for half a second: after seconds of host-side generation the GPU is in a low-power state, and the first tens of
# milliseconds run below the steady clocks (3.13 vs 2.83 ms per launch on rndm * 4096; profiles/r02_clock_power.txt)
import time
t0 = time.time()
while time.time() - t0 < float(os.environ.get("VKMR_PROBE_WARM_S", "0.5")):
    for _ in range(4):
        dev.map_async(d_data, b.words, d_meta, b.count, d_out)
    dev.sync()
for e0, e1 in ev:
    dev.record(e0)
    dev.map_async(d_data, b.words, d_meta, b.count, d_out)
    dev.record(e1)
dev.sync()
ms = [dev.elapsed_ms(e0, e1) for e0, e1 in ev]
chk = int(dev.download(d_out, 32 * b.count).astype(np.uint64).sum())
print(f"2^{log2} x rndm {maxlen}", {k: v for k, v in os.environ.items() if k.startswith("VKMR_MAP")}, "ms per launch", round(float(np.mean(ms)), 4), "min", round(min(ms), 4), "digest checksum", chk)
