"""One slice of 2^k random digests reduced N times (for rocprofv3 --kernel-trace --stats: per-kernel times of the reduction's
launch sequence, e.g. under VKMR_HIP_LIB=<another build>).  GPU box only:  python3 tools/reduce_probe.py [26] [20]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import vk_merkle_roots_amd as vk  # noqa: E402

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 26
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = vk.HipDevice(0)
n = 1 << log2
rng = np.random.default_rng(1)
d = dev.upload(rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32))
scr, root = dev.reduce_scratch(n), dev.alloc(32)
e0, e1 = dev.new_event(), dev.new_event()
for _ in range(3):
    dev.reduce_async(d, n, log2, scr, root)
dev.sync()
dev.record(e0)
for _ in range(reps):
    dev.reduce_async(d, n, log2, scr, root)
dev.record(e1)
dev.sync()
print(f"reduce 2^{log2}: {dev.elapsed_ms(e0, e1) / reps:.4f} ms per reduction; root", vk.engine.digest_hex(dev.download(root, 32)))
