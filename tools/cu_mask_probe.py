#!/usr/bin/env python3
"""Does a CU share something with its neighbour that the hash kernels are short of (the instruction cache is one per two CUs)?
Step 1: where does bit i of a stream's CU mask land?  tools/where.hip stores XCC_ID and HW_ID per workgroup; every bit is tried alone.
Step 2: the map kernel and the bulk reduce on streams whose masks enable HALF the CUs of every shader engine, chosen by PHYSICAL CU id:
even ids (one CU of each neighbouring pair), ids with bit 1 clear (pairs kept together), the lower half.  If neighbours compete for
something, "even ids" is faster than the patterns that keep neighbours together.
GPU box:  hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/libwhere.so tools/where.hip && python3 tools/cu_mask_probe.py [--log2 24]"""
import argparse
import ctypes as C
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
hip = C.CDLL("libamdhip64.so")
n = 1 << a.log2
b = vk.rndm_packed(42, n, 127)
d_data, d_meta, d_out = dev.upload(b.data), dev.upload(b.meta), dev.alloc(32 * n)
d_scr, d_root = dev.reduce_scratch(n), dev.alloc(32)
height = vk.tree_height(n)


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << k for k in range(32) if bits[32 * w + k]) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return s.value


where = C.CDLL(os.path.join(ROOT, "tools", "libwhere.so"))
where.where_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
NWG = 16384
d_where = dev.alloc(8 * NWG)


def placement(stream):
    """The set of (xcc, se, sh, cu) that workgroups of a launch on `stream` ran on."""
    rc = where.where_launch(stream, d_where.ptr, NWG)
    if rc != 0:
        raise RuntimeError(f"where_launch: {rc}")
    dev.sync(stream)
    w = dev.download(d_where, 8 * NWG).view(np.uint32).reshape(-1, 2)
    xcc, hw = w[:, 0] & 15, w[:, 1]
    return sorted(set(zip(xcc.tolist(), ((hw >> 13) & 7).tolist(), ((hw >> 12) & 1).tolist(), ((hw >> 8) & 15).tolist())))


def destroy(stream):
    hip.hipStreamDestroy(C.c_void_p(stream))


s_all = masked_stream([1] * 256)
everywhere = placement(s_all)
destroy(s_all)
print(f"all bits set: {len(everywhere)} distinct (xcc, se, sh, cu)")
for x in sorted({p[0] for p in everywhere}):
    print(f"  xcc {x}: " + "  ".join(f"se{se}.sh{sh}:{[p[3] for p in everywhere if p[:3] == (x, se, sh)]}"
                                     for se, sh in sorted({p[1:3] for p in everywhere if p[0] == x})))
# Bit i of the mask: XCC i % 8, then k = i // 8 counts that XCC's ACTIVE CUs shader engine first: se = k % 4, the (k // 4)-th active CU
# of that engine.  An XCC whose bits are all zero runs UNRESTRICTED (so "every other bit" masks nothing).  Checked on a few bits:
active = {}
for p_ in everywhere:
    active.setdefault(p_[:3], []).append(p_[3])
for i in (0, 8, 16, 24, 32, 93, 208, 255):
    s = masked_stream([1 if k == i else 0 for k in range(256)])
    got = [p_ for p_ in placement(s) if p_[0] == i % 8]
    destroy(s)
    k = i // 8
    want = (i % 8, k % 4, 0, sorted(active[(i % 8, k % 4, 0)])[k // 4])
    print(f"bit {i:3d} alone -> on its XCC {got}; expected {want}: {'ok' if got == [want] else 'MISMATCH'}")


def choose(pick):
    """Mask enabling, in every shader engine of every XCC, the CUs pick(sorted active ids) returns (always four: workgroups go round the
    engines evenly, so the engine with the fewest CUs sets the time)."""
    bits = [0] * 256
    for (xcc, se, sh), ids in active.items():
        ids = sorted(ids)
        for cu in pick(ids):
            bits[xcc + 8 * (se + 4 * ids.index(cu))] = 1
    return bits


def one_per_pair(ids):      # four CUs from four different (2j, 2j+1) pairs: no enabled CU has an enabled neighbour
    out, seen = [], set()
    for cu in ids:
        if cu >> 1 not in seen:
            seen.add(cu >> 1)
            out.append(cu)
    return out[:4]


def whole_pairs(ids):       # two pairs with both CUs alive: every enabled CU has its neighbour enabled
    pairs = [j for j in range(5) if 2 * j in ids and 2 * j + 1 in ids]
    return [2 * pairs[0], 2 * pairs[0] + 1, 2 * pairs[1], 2 * pairs[1] + 1]


# Neighbours (2j, 2j+1) share an instruction cache and a scalar cache.
patterns = {
    "all 256 CUs": [1] * 256,
    "4 per engine, no two neighbours": choose(one_per_pair),
    "4 per engine, two whole pairs": choose(whole_pairs),
    "4 per engine, the first four alive": choose(lambda ids: ids[:4]),
}
e0, e1 = dev.new_event(), dev.new_event()
for name, bits in patterns.items():
    s = masked_stream(bits)
    ncu = len(placement(s))
    res = {}
    for what, fn in (("map", lambda: dev.map_async(d_data, b.words, d_meta, n, d_out, stream=s)),
                     ("reduce", lambda: dev.reduce_async(d_out, n, height, d_scr, d_root, stream=s))):
        for _ in range(6):
            fn()
        dev.sync(s)
        t = []
        for _ in range(8):
            dev.record(e0, s); fn(); dev.record(e1, s); dev.sync(s)
            t.append(dev.elapsed_ms(e0, e1))
        res[what] = float(np.median(t))
    print(f"{name:46s} map {res['map']:.3f} ms  reduce {res['reduce']:.3f} ms   ({sum(bits)} bits, {ncu} CUs seen)")
