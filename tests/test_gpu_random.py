"""Randomised GPU sweeps against the oracle (fixed seeds): reductions of random counts/heights, single and
batched; map batches with adversarial length mixes (padding boundaries, empty strings, a few long ones)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_random_reductions(gpu, oracle):
    rng = np.random.default_rng(2024)
    for case in range(120):
        n = int(rng.choice([rng.integers(1, 300), rng.integers(1, 5000), rng.integers(1, 300000)]))
        need = max(1, int(n - 1).bit_length())
        height = need + int(rng.choice([0, 0, 0, 1, 3, 9]))
        leaves = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32)
        got = gpu.reduce_digests(leaves, height=height)
        assert (got == oracle.reduce_height(leaves, height)).all(), (case, n, height)


def test_random_batched_reductions(gpu, oracle):
    rng = np.random.default_rng(2025)
    for case in range(40):
        cap_log2 = int(rng.integers(1, 16))
        cap = 1 << cap_log2
        nslices = int(rng.integers(1, 40))
        last = int(rng.integers(1, cap + 1))
        n = (nslices - 1) * cap + last
        leaves = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32)
        height = cap_log2 if nslices > 1 else max(1, int(last - 1).bit_length())
        d_in = gpu.upload(leaves)
        d_scratch = gpu.alloc(gpu.lib.vkmr_hip_reduce_slices_scratch_bytes(cap, nslices))
        d_roots = gpu.alloc(32 * nslices)
        gpu.reduce_slices_async(d_in, nslices, cap, last, height, d_scratch, d_roots)
        got = gpu.download(d_roots, 32 * nslices).reshape(-1, 8)
        for k in range(nslices):
            want = oracle.reduce_height(leaves[k * cap: min(n, (k + 1) * cap)], height)
            assert (got[k] == want).all(), (case, cap_log2, nslices, last, k)
        for b in (d_in, d_scratch, d_roots):
            b.free()


def _batch(strings):
    import vk_merkle_roots_amd as vk
    meta = np.zeros((len(strings), 2), dtype=np.uint32)
    chunks, w = [], 0
    for i, s in enumerate(strings):
        meta[i] = (w, len(s))
        nw = (len(s) + 3) // 4
        chunks.append(s + b"\0" * (4 * nw - len(s)))
        w += nw
    data = np.frombuffer(b"".join(chunks), dtype=np.uint8).view(np.uint32).copy() if w else np.zeros(0, np.uint32)
    return vk.PackedBatch(data, meta, w, sum(len(s) for s in strings))


def test_random_map_length_mixes(gpu, oracle):
    rng = np.random.default_rng(2026)
    boundary = [0, 1, 3, 4, 55, 56, 57, 63, 64, 65, 119, 120, 121, 127, 128, 183, 184, 247, 248]
    for case in range(30):
        n = int(rng.integers(1, 6000))
        kind = case % 5
        if kind == 0:
            lens = rng.choice(boundary, size=n)
        elif kind == 1:
            lens = rng.integers(0, 130, size=n)
        elif kind == 2:
            lens = np.full(n, int(rng.choice(boundary)))
        elif kind == 3:
            lens = np.where(rng.random(n) < 0.02, rng.integers(1000, 9000, size=n), rng.integers(0, 60, size=n))
        else:
            lens = rng.integers(0, 1500, size=n)
        blob = rng.integers(0, 256, size=int(lens.sum()) + 1, dtype=np.uint8).tobytes()
        strings, pos = [], 0
        for ln in lens:
            strings.append(blob[pos: pos + int(ln)])
            pos += int(ln)
        b = _batch(strings)
        got = gpu.leaf_digests(b)
        want = oracle.leaves_packed(b.data, b.meta, threads=16)
        assert (got == want).all(), (case, kind, n)


@pytest.mark.parametrize("variant", ["0", "1", "2", "3", "4", "5", "9", "10", "20", "21", "23", "24", "26", "27"])
def test_every_fetch_mode_of_the_map_kernel_is_bit_exact(variant):
    """The map kernel's fetch modes against the oracle on short, long, ragged, unordered and out-of-range inputs --
    whichever mode a launch picks, the digests are the same.  "0" is the product library choosing from the batch alone;
    the others force one mode through the EXPERIMENTS build (build/ab/libexp.so, VKMR_MAP_VARIANT read once per process:
    LDS-staged 64 / 32 KiB tiles, per-wavefront gather, per-lane 16-byte loads, line window, and the north star's
    K-in-LDS / schedule-ring-in-LDS forms of the compression)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import vk_merkle_roots_amd as vk
from conftest import Oracle, golden_pattern
from test_gpu_parity import batch_of
o, gpu = Oracle(), vk.HipDevice(0)
for seed, n, maxlen in ((3, 70000, 127), (5, 3000, 4096), (6, 9000, 700), (7, 4000, 2), (8, 150000, 300)):
    b = vk.rndm_packed(seed, n, maxlen)
    assert (gpu.leaf_digests(b) == o.leaves_packed(b.data, b.meta, threads=8)).all(), (seed, n, maxlen)
b = batch_of([golden_pattern(k, 9 + k) for k in range(0, 400)] + [golden_pattern(k, k) for k in (4095, 4096, 4097, 8191, 70000)])
assert (gpu.leaf_digests(b) == o.leaves_packed(b.data, b.meta)).all()
b = vk.rndm_packed(31, 6000, 1500)
perm = np.random.default_rng(1).permutation(6000)
meta = b.meta[perm].copy()
meta[5] = (b.words - 1, 64); meta[6] = (b.words + 99, 7); meta[7] = (b.words - 3, 0xFFFFFFF0)
cut = meta.copy(); cut[5, 1] = 4; cut[6] = (0, 0); cut[7, 1] = 12
got = gpu.leaf_digests(vk.PackedBatch(b.data, meta, b.words, 0))
assert (got == o.leaves_packed(b.data, cut)).all()
# a batch that does not start on a 128-byte boundary (sub-buffer of a larger allocation)
d_all = gpu.upload(np.concatenate([np.zeros(5, np.uint32), b.data]))
d_meta, d_out = gpu.upload(b.meta), gpu.alloc(32 * b.count)
vk.check(gpu.lib.vkmr_hip_map_async(gpu.index, gpu.stream, d_all.at(20), b.words, d_meta.ptr, b.count, d_out.ptr), "map")
assert (gpu.download(d_out, 32 * b.count).reshape(-1, 8) == o.leaves_packed(b.data, b.meta, threads=8)).all()
print("ok")
''' % (ROOT, ROOT)
    env = dict(os.environ)
    env.pop("VKMR_HIP_LIB", None)
    if variant != "0":
        from vk_merkle_roots_amd.build import EXP_LIB
        assert os.path.exists(EXP_LIB), "experiments library not built (vk_merkle_roots_amd.build.build_experiments)"
        env.update(VKMR_MAP_VARIANT=variant, VKMR_HIP_LIB=EXP_LIB)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stderr[-2000:].decode()
