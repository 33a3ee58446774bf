"""Is the long-string map kernel held back by memory?  Same 2^21 strings of rndm 42 * 4096 (same sizes, same
block counts, same tile order), once with their real starts (4.3 GB streamed) and once with every start folded
into the first 1 MiB of the batch (everything L2-resident): the difference is what memory costs.  GPU box only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import vk_merkle_roots_amd as vk  # noqa: E402

dev = vk.HipDevice(0)
b = vk.rndm_packed(42, 1 << 21, 4096)
folded = b.meta.copy()
folded[:, 0] = folded[:, 0] % np.uint32(1 << 18)          # 1 MiB window; sizes <= 4095 B stay inside the 4.3 GB buffer
d_data, d_out = dev.upload(b.data), dev.alloc(32 * b.count)
for name, meta in (("streamed (real starts)", b.meta), ("L2-resident (starts folded into 1 MiB)", folded)):
    d_meta = dev.upload(meta)
    ev = [(dev.new_event(), dev.new_event()) for _ in range(5)]
    dev.map_async(d_data, b.words, d_meta, b.count, d_out)
    dev.sync()
    for e0, e1 in ev:
        dev.record(e0)
        dev.map_async(d_data, b.words, d_meta, b.count, d_out)
        dev.record(e1)
    dev.sync()
    ms = [dev.elapsed_ms(e0, e1) for e0, e1 in ev]
    print(f"{name}: {np.mean(ms):.4f} ms per launch (min {min(ms):.4f})")
