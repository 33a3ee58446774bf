"""BASELINE.json's full sizes on the GPU: configs[2] (rndm 2^26 x 127 B) against the
threaded oracle and through size-independent properties (any slicing gives the same
root; both reduction variants agree), and configs[4] in miniature-at-scale (2^18 x 4 KiB
strings: the multi-block padding path)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def gpu_digests(gpu, batch, batch_strings):
    d_out = gpu.alloc(32 * batch.count)
    for b0 in range(0, batch.count, batch_strings):
        sub = batch.slice(b0, min(batch.count, b0 + batch_strings))
        d_data, d_meta = gpu.upload(sub.data), gpu.upload(sub.meta)
        gpu.map_async(d_data, sub.words, d_meta, sub.count, d_out, out_offset_digests=b0)
        gpu.sync()
        d_data.free()
        d_meta.free()
    return d_out


def test_config3_2p26_leaves(gpu, oracle):
    import vk_merkle_roots_amd as vk
    from vk_merkle_roots_amd.engine import digest_hex, tree_height
    n = 1 << 26
    batch = vk.rndm_packed(42, n, 127)
    assert batch.count == n
    d_digests = gpu_digests(gpu, batch, 1 << 23)

    # (1) the oracle, threaded, on the same packed input: every leaf digest and the root
    want_leaves = oracle.leaves_packed(batch.data, batch.meta, threads=64)
    got = gpu.download(d_digests, 32 * n).reshape(-1, 8)
    assert (got == want_leaves).all()
    del got
    want_root = oracle.hex(oracle.root(want_leaves, threads=64))
    del want_leaves
    # ... and what the reference's own CPU path printed for this stream (tests/golden/big_roots.json)
    import json
    import os
    from conftest import ROOT
    big = json.load(open(os.path.join(ROOT, "tests", "golden", "big_roots.json")))
    assert big["count"] == n and want_root == big["sub_roots"]["42"]["root"]

    # (2) 8 slices of 2^23 reduced by one batched call + on-device combine
    d_scratch = gpu.alloc(gpu.lib.vkmr_hip_reduce_slices_scratch_bytes(1 << 23, 8))
    d_roots = gpu.alloc(32 * 8)
    gpu.reduce_slices_async(d_digests, 8, 1 << 23, 1 << 23, 23, d_scratch, d_roots)
    d_top, d_final = gpu.reduce_scratch(8), gpu.alloc(32)
    gpu.reduce_async(d_roots, 8, 3, d_top, d_final)
    root_8 = digest_hex(gpu.download(d_final, 32))
    assert root_8 == want_root

    # (3) slicing does not matter: one slice of 2^26, and 64 slices of 2^20, same root
    d_s1 = gpu.reduce_scratch(n)
    gpu.reduce_async(d_digests, n, 26, d_s1, d_final)
    assert digest_hex(gpu.download(d_final, 32)) == want_root
    d_s64 = gpu.alloc(gpu.lib.vkmr_hip_reduce_slices_scratch_bytes(1 << 20, 64))
    d_r64 = gpu.alloc(32 * 64)
    gpu.reduce_slices_async(d_digests, 64, 1 << 20, 1 << 20, 20, d_s64, d_r64)
    gpu.reduce_async(d_r64, 64, 6, d_top, d_final)
    assert digest_hex(gpu.download(d_final, 32)) == want_root

    # (4) ragged: drop the last 12345 leaves -> last slice short, reduced to capacity height
    m = n - 12345
    gpu.reduce_slices_async(d_digests, 8, 1 << 23, m - 7 * (1 << 23), 23, d_scratch, d_roots)
    gpu.reduce_async(d_roots, 8, 3, d_top, d_final)
    ragged_sliced = digest_hex(gpu.download(d_final, 32))
    gpu.reduce_async(d_digests, m, tree_height(m), d_s1, d_final)
    assert digest_hex(gpu.download(d_final, 32)) == ragged_sliced


def test_config5_long_strings(gpu, oracle):
    """rndm <seed> 2^18 4096: lengths 1..4095, 1..65 blocks per string (BASELINE configs[4] at 1/64 of its count)."""
    import vk_merkle_roots_amd as vk
    n = 1 << 18
    batch = vk.rndm_packed(11, n, 4096)
    d_digests = gpu_digests(gpu, batch, 1 << 16)
    got = gpu.download(d_digests, 32 * n).reshape(-1, 8)
    want = oracle.leaves_packed(batch.data, batch.meta, threads=64)
    assert (got == want).all()
    assert vk.merkle_root_packed_batched(gpu, batch, slice_capacity=1 << 16, batch_strings=1 << 15) == oracle.hex(oracle.root(want, threads=64))


def test_config5_full_2p24_x_4k(gpu, oracle):
    """BASELINE configs[4] at full size: rndm <seed> 2^24 4096 (about 34 GB of input, 1..65 blocks per
    string), streamed as 16 batches of 2^20 strings (a packed batch addresses at most 2^32 words) into 2
    slices of 2^23 (the reference's slice); every leaf digest and the root against the threaded oracle -- and the root
    against what the REFERENCE's CPU path printed for this very stream (tests/golden/big_roots.json, config5: rndm 42 2^24 4096,
    38 minutes of the reference)."""
    import vk_merkle_roots_amd as vk
    from vk_merkle_roots_amd.engine import digest_hex
    n, per = 1 << 24, 1 << 20
    stream = vk.RndmStream(42, 4096)
    d_digests = gpu.alloc(32 * n)
    want = np.zeros((n, 8), dtype=np.uint32)
    total_bytes = 0
    for b0 in range(0, n, per):
        batch = stream.next(per)
        total_bytes += batch.nbytes
        d_data, d_meta = gpu.upload(batch.data), gpu.upload(batch.meta)
        gpu.map_async(d_data, batch.words, d_meta, per, d_digests, out_offset_digests=b0)
        want[b0: b0 + per] = oracle.leaves_packed(batch.data, batch.meta, threads=64)   # CPU checks while the GPU maps
        gpu.sync()
        d_data.free()
        d_meta.free()
    stream.close()
    assert total_bytes > 30e9
    got = gpu.download(d_digests, 32 * n).reshape(-1, 8)
    assert (got == want).all()
    del got
    d_scratch = gpu.alloc(gpu.lib.vkmr_hip_reduce_slices_scratch_bytes(1 << 23, 2))
    d_roots, d_top, d_final = gpu.alloc(64), gpu.reduce_scratch(2), gpu.alloc(32)
    gpu.reduce_slices_async(d_digests, 2, 1 << 23, 1 << 23, 23, d_scratch, d_roots)
    gpu.reduce_async(d_roots, 2, 1, d_top, d_final)
    root = digest_hex(gpu.download(d_final, 32))
    assert root == oracle.hex(oracle.root(want, threads=64))
    import json
    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_roots.json")))["config5"]
    assert golden["generator"] == "rndm 42 16777216 4096" and total_bytes == golden["bytes"] and root == golden["root"]


def test_one_string_of_512_mib_plus_5_bytes(gpu, oracle):
    """The 64-bit bit length: a string of 2^29 + 5 bytes has bit length 2^32 + 40, so the high length word of the
    padding (size >> 29) is 1 -- zero in every other test.  The reference's own shader is wrong here
    (MB_SIZE_TOP, src/common/SHA-256defs.h:30, masks instead of shifting), so the CPU path (the oracle) is the
    authority.  One lane hashes 8.4 M blocks in sequence -- SHA-256 is a chain -- so this one launch takes
    the better part of a minute; two short strings ride along in the same batch."""
    import vk_merkle_roots_amd as vk
    big = (1 << 29) + 5
    rng = np.random.default_rng(2929)
    words = (big + 3) // 4
    data = rng.integers(0, 2**32, size=words + 32, dtype=np.uint32)
    raw = data.view(np.uint8)
    raw[big: 4 * words] = 0
    small = [(words, 3), (words + 1, 100)]
    meta = np.array([(0, big)] + small, dtype=np.uint32)
    batch = vk.PackedBatch(data, meta, words + 32, big + 103)
    got = gpu.leaf_digests(batch)
    want = oracle.leaves_packed(data, meta, threads=3)
    assert (got == want).all()
    # and the oracle itself against hashlib on this one input (the golden vectors stop at 4 KiB)
    import hashlib
    h = hashlib.sha256(hashlib.sha256(raw[:big].tobytes()).digest()).digest()
    assert oracle.hex(want[0]) == h.hex()
