#!/usr/bin/env python3
"""Which strings of a batch does a map-kernel variant get wrong?  (GPU box; VKMR_HIP_LIB / VKMR_MAP_VARIANT in the environment.)
    python3 tools/diag_variant.py seed count maxlen [seed count maxlen ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vk_merkle_roots_amd as vk  # noqa: E402
from conftest import Oracle  # noqa: E402

o, gpu = Oracle(), vk.HipDevice(0)
args = [int(x) for x in sys.argv[1:]]
for k in range(0, len(args), 3):
    seed, n, maxlen = args[k:k + 3]
    b = vk.rndm_packed(seed, n, maxlen)
    got = gpu.leaf_digests(b)
    want = o.leaves_packed(b.data, b.meta, threads=8)
    bad = np.nonzero((got != want).any(axis=1))[0]
    info = gpu.lib.vkmr_hip_kernel_info().decode().split(" reduce=")[0]
    print(f"rndm {seed} {n} {maxlen}: {len(bad)} of {n} wrong; {info}")
    if len(bad):
        sizes = b.meta[bad, 1]
        print("  first wrong indices:", bad[:12].tolist(), "sizes:", sizes[:12].tolist(), "blocks:", ((sizes[:12].astype(np.int64) + 8) // 64 + 1).tolist())
        print("  wrong by block count:", dict(zip(*np.unique((sizes.astype(np.int64) + 8) // 64 + 1, return_counts=True))))
        allb = (b.meta[:, 1].astype(np.int64) + 8) // 64 + 1
        print("  all by block count  :", dict(zip(*np.unique(allb, return_counts=True))) if n <= 5000 else "(many)")
        starts = b.meta[bad, 0].astype(np.int64)
        print("  wrong strings that end within 16 words of the buffer's end:", int(((starts + (sizes + 3) // 4 + 16) > b.words).sum()))
        z = np.nonzero((got[bad] == 0).all(axis=1))[0]
        print("  wrong digests that are all zero (never written):", len(z))
