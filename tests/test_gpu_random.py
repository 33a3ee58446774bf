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
