"""GPU parity: the HIP path, called through the C ABI, against the oracle and the
golden vectors.  Bit-exact (integer work).  Run with -m gpu on an MI355X."""
import numpy as np
import pytest

from conftest import golden_pattern

pytestmark = pytest.mark.gpu


def batch_of(strings):
    import vk_merkle_roots_amd as vk
    meta = np.zeros((len(strings), 2), dtype=np.uint32)
    words = []
    w = 0
    for i, s in enumerate(strings):
        meta[i] = (w, len(s))
        nw = (len(s) + 3) // 4
        words.append(s + b"\0" * (4 * nw - len(s)))
        w += nw
    data = np.frombuffer(b"".join(words), dtype=np.uint32).copy() if w else np.zeros(0, np.uint32)
    return vk.PackedBatch(data, meta, w, sum(len(s) for s in strings))


# ---- map ------------------------------------------------------------------------

def test_metadata_from_sizes(gpu):
    """vkmr_hip_metadata_from_sizes_async: entry i = {first_word + sum over j < i of ceil(size[j] / 4), size[j]} -- what
    Batch::Push computes on the host (reference src/vkmr/Batches.cpp:64-121) -- against numpy, for counts around the
    kernels' 16-per-lane and 4096-per-workgroup edges, a batch of 2^20 + 3 strings, zero sizes and the largest size."""
    import vk_merkle_roots_amd as vk
    rng = np.random.default_rng(77)
    for count, first_word, hi in ((1, 0, 200), (15, 7, 200), (16, 0, 200), (17, 3, 200), (4095, 0, 128), (4096, 1, 128), (4097, 0, 128),
                                  (100000, 12345, 65535), ((1 << 20) + 3, 0, 128), (300000, 0, 1)):
        sizes = rng.integers(0, hi + 1, size=count, dtype=np.uint32).astype(np.uint16)
        if hi == 65535:
            sizes[[0, count // 2, count - 1]] = 65535
        d_sizes = gpu.upload(sizes)
        d_meta = gpu.alloc(8 * count + 16)
        d_scratch = gpu.alloc(gpu.lib.vkmr_hip_sizes_scratch_bytes(count))
        vk.check(gpu.lib.vkmr_hip_metadata_from_sizes_async(gpu.index, gpu.stream, d_sizes.ptr, count, first_word, d_scratch.ptr, d_meta.ptr), "from_sizes")
        got = gpu.download(d_meta, 8 * count).reshape(-1, 2)
        words = (sizes.astype(np.uint64) + 3) // 4
        starts = first_word + np.concatenate(([0], np.cumsum(words)[:-1]))
        assert np.array_equal(got[:, 1], sizes.astype(np.uint32)), count
        assert np.array_equal(got[:, 0].astype(np.uint64), starts.astype(np.uint64)), count
        for b in (d_sizes, d_meta, d_scratch):
            b.free()
    assert gpu.lib.vkmr_hip_metadata_from_sizes_async(gpu.index, gpu.stream, None, 0, 0, None, None) == 0      # nothing to do
    assert gpu.lib.vkmr_hip_metadata_from_sizes_async(gpu.index, gpu.stream, None, 5, 0, None, None) < 0       # null pointers are refused


def _split_on_gpu(gpu, text, meta_capacity=None, data_capacity_words=None):
    import vk_merkle_roots_amd as vk
    from vk_merkle_roots_amd import _abi
    exp = _abi.experiments_lib()      # the splitter lives in the experiments build since round 4 (include/vkmr_hip_experiments.h)
    n = len(text)
    lines = text.count(b"\n")
    meta_capacity = meta_capacity if meta_capacity is not None else max(1, lines)
    data_capacity_words = data_capacity_words if data_capacity_words is not None else n // 4 + lines + 4
    padded = np.frombuffer(text + b"\xAA" * (48 - n % 16), dtype=np.uint8)      # readable (and not zero) behind the text
    d_text = gpu.upload(padded)
    d_data = gpu.alloc(4 * data_capacity_words + 64)
    d_meta = gpu.alloc(8 * meta_capacity + 16)
    d_scratch = gpu.alloc(exp.vkmr_hip_split_scratch_bytes(n, meta_capacity))
    d_result = gpu.alloc(16)
    vk.check(gpu.lib.vkmr_hip_memset_async(gpu.index, gpu.stream, d_data.ptr, 0xCD, 4 * data_capacity_words + 64), "memset")
    vk.check(exp.vkmr_hip_split_text_async(gpu.index, gpu.stream, d_text.ptr, n, d_scratch.ptr, d_data.ptr, data_capacity_words, d_meta.ptr, meta_capacity,
                                               d_result.ptr), "split")
    strings, words, status = (int(x) for x in gpu.download(d_result, 12))
    meta = gpu.download(d_meta, 8 * min(strings, meta_capacity)).reshape(-1, 2) if strings else np.zeros((0, 2), np.uint32)
    data = gpu.download(d_data, 4 * data_capacity_words + 64)
    for b in (d_text, d_data, d_meta, d_scratch, d_result):
        b.free()
    return strings, words, status, meta, data


def test_split_text_on_the_device(gpu):
    """vkmr_hip_split_text_async against the host packer (the memchr-and-memcpy form, which the oracle's line rules pin):
    same strings, same entries, same packed words with zero padding, nothing written behind them -- for rndm-like text,
    one- and two-byte lines, empty lines in runs and at both ends, CRs, lines that straddle the kernels' 16-byte pieces and
    4 KiB blocks, lines of 5 000 and 70 000 bytes, a text of one newline, and 40 MB of text."""
    import vk_merkle_roots_amd as vk
    h = vk.host_lib()
    rng = np.random.default_rng(123)

    def rand_lines(count, lo, hi, alphabet=(33, 126)):
        return b"".join(rng.integers(alphabet[0], alphabet[1] + 1, size=int(k), dtype=np.uint8).tobytes() + b"\n" for k in rng.integers(lo, hi + 1, size=count))

    texts = [b"\n", b"a\n", b"\n\n\nab\n\n", b"x" * 15 + b"\n" + b"y" * 16 + b"\n" + b"z" * 17 + b"\n", b"q\r\n\r\n",
             rand_lines(5000, 0, 127), rand_lines(20000, 0, 3), rand_lines(3000, 1, 1), rand_lines(300, 4000, 5000),
             b"head\n" + b"L" * 70000 + b"\n" + rand_lines(100, 0, 40) + b"M" * 4095 + b"\n" + b"N" * 4096 + b"\n" + b"O" * 4097 + b"\n",
             rand_lines(50, 0, 20, alphabet=(9, 13)),                          # tabs, newlines and CRs only: runs of empty lines
             rand_lines(600000, 0, 127)]
    for text in texts:
        buf = np.frombuffer(text, dtype=np.uint8)
        lines = text.count(b"\n")
        ref_data = np.zeros(len(text) // 4 + lines + 4, dtype=np.uint32)
        ref_meta = np.zeros((lines + 1, 2), dtype=np.uint32)
        wu, bt = C_u64(), C_u64()
        want = h.vkmr_host_pack_lines_portable(buf.ctypes.data, len(text), ref_data.ctypes.data, len(ref_data), ref_meta.ctypes.data, len(ref_meta), wu, bt)
        strings, words, status, meta, data = _split_on_gpu(gpu, text)
        assert (strings, words, status) == (want, wu.value, 0), (len(text), strings, want)
        assert np.array_equal(meta, ref_meta[:want])
        assert np.array_equal(data[:words], ref_data[:words])
        assert (data[words:] == 0xCDCDCDCD).all()                              # nothing written behind the strings' words
    # what does not fit is reported, not written past the capacities
    text = rand_lines(1000, 10, 50)
    strings, words, status, meta, data = _split_on_gpu(gpu, text, meta_capacity=999)
    assert status == 1
    strings, words, status, meta, data = _split_on_gpu(gpu, text, data_capacity_words=100)
    assert status == 1 and strings == 1000 and (data == 0xCDCDCDCD).all()
    # an unterminated tail is not a string (the caller appends the newline): "ab\ncd" holds one
    strings, words, status, meta, data = _split_on_gpu(gpu, b"ab\ncd")
    assert (strings, words, status) == (1, 1, 0)


def C_u64():
    import ctypes
    return ctypes.c_uint64()


def test_warm_up_leaves_no_trace(gpu, oracle):
    """vkmr_hip_warm_up (kernels, copy engine, both, neither; small and large copies; the device's stream and a new one)
    returns VKMR_OK, and launches after it give what they give without it."""
    import vk_merkle_roots_amd as vk
    b = vk.rndm_packed(5, 3000, 127)
    want = [oracle.hex(r) for r in gpu.leaf_digests(b)]
    other = gpu.new_stream()
    for kernels, copy_bytes, stream in ((True, 1 << 20, None), (True, 0, None), (False, 256, None), (False, 0, None), (True, 40 << 20, other), (True, 1, other)):
        gpu.warm_up(kernels=kernels, copy_bytes=copy_bytes, stream=stream)
        assert [oracle.hex(r) for r in gpu.leaf_digests(b)] == want

def test_map_golden_leaves(gpu, oracle, golden):
    msgs = [golden_pattern(v["len"], v["salt"]) for v in golden["leaves"]]
    got = gpu.leaf_digests(batch_of(msgs))
    for v, row in zip(golden["leaves"], got):
        assert oracle.hex(row) == v["sha256d"], v["len"]


def test_map_every_length_0_to_300(gpu, oracle):
    msgs = [golden_pattern(n, 5 + n) for n in range(0, 301)]
    b = batch_of(msgs)
    got = gpu.leaf_digests(b)
    want = oracle.leaves_packed(b.data, b.meta)
    assert (got == want).all()


def test_map_masks_stale_tail_bytes(gpu, oracle):
    """bytes after `size` in the last word must not reach the hash."""
    msgs = [golden_pattern(n, n) for n in (1, 2, 3, 5, 6, 7, 57, 62, 63, 121)]
    b = batch_of(msgs)
    dirty = b.data.copy()
    raw = dirty.view(np.uint8)
    for st, sz in b.meta:
        for k in range(int(sz), 4 * ((int(sz) + 3) // 4)):
            raw[4 * int(st) + k] = 0xEE
    import vk_merkle_roots_amd as vk
    got = gpu.leaf_digests(vk.PackedBatch(dirty, b.meta, b.words, b.nbytes))
    want = oracle.leaves_packed(b.data, b.meta)
    assert (got == want).all()


def test_map_last_thread_is_bounded(gpu, oracle):
    """a count that is not a multiple of the workgroup must not write past it."""
    import vk_merkle_roots_amd as vk
    b = vk.rndm_packed(21, 1000, 127)
    d_data, d_meta = gpu.upload(b.data), gpu.upload(b.meta)
    sentinel = np.full((1024, 8), 0xA5A5A5A5, dtype=np.uint32)
    d_out = gpu.upload(sentinel)
    gpu.map_async(d_data, b.words, d_meta, 1000, d_out)
    out = gpu.download(d_out, 1024 * 32).reshape(-1, 8)
    assert (out[:1000] == oracle.leaves_packed(b.data, b.meta)).all()
    assert (out[1000:] == 0xA5A5A5A5).all()


@pytest.mark.parametrize("seed,count,maxlen", [(1712489279, 1024, 127), (7, 1000, 300), (42, 4096, 4096), (3, 50000, 127),
                                               (4, 3000, 2), (5, 777, 65), (6, 2000, 1024)])
def test_map_rndm_batches(gpu, oracle, seed, count, maxlen):
    import vk_merkle_roots_amd as vk
    b = vk.rndm_packed(seed, count, maxlen)
    got = gpu.leaf_digests(b)
    want = oracle.leaves_packed(b.data, b.meta, threads=8)
    assert (got == want).all()


def test_map_metadata_in_any_order(gpu, oracle):
    """The ABI takes any metadata, not only Batch::Push's ascending layout: permuted entries,
    entries sharing bytes, a tile whose strings are far apart (falls off the staged path)."""
    import vk_merkle_roots_amd as vk
    b = vk.rndm_packed(31, 5000, 127)
    rng = np.random.default_rng(9)
    perm = rng.permutation(5000)
    meta = b.meta[perm].copy()
    meta = np.concatenate([meta, meta[:100]])            # duplicates: same bytes hashed twice
    shuffled = vk.PackedBatch(b.data, meta, b.words, 0)
    got = gpu.leaf_digests(shuffled)
    want = oracle.leaves_packed(b.data, meta)
    assert (got == want).all()


def test_map_out_of_range_metadata_does_not_fault(gpu, oracle):
    """A start/size pointing past data_words must not read out of bounds nor spin: the string is cut
    at the end of the buffer (include/vkmr_hip.h, vkmr_hip_map_async)."""
    import vk_merkle_roots_amd as vk
    b = vk.rndm_packed(32, 1000, 127)
    meta = b.meta.copy()
    meta[10] = (b.words - 1, 64)            # runs 15 words past the end -> 4 bytes left
    meta[20] = (b.words + 1000, 5)          # entirely outside -> empty string
    meta[30] = (b.words - 2, 0xFFFFFFF0)    # 4 GiB claimed, 8 bytes left
    meta[999] = (b.words - 2, 200)
    got = gpu.leaf_digests(vk.PackedBatch(b.data, meta, b.words, 0))
    cut = meta.copy()
    cut[10, 1], cut[20], cut[30, 1], cut[999, 1] = 4, (0, 0), 8, 8
    want = oracle.leaves_packed(b.data, cut)
    assert (got == want).all()


def test_map_into_sub_slice_offsets(gpu, oracle):
    """Two batches mapped into one slice at different offsets (Slice::Sub, reference Slices.h:145-187)."""
    import vk_merkle_roots_amd as vk
    b = vk.rndm_packed(8, 3000, 127)
    first, second = b.slice(0, 1234), b.slice(1234, 3000)
    d_out = gpu.alloc(32 * 3000)
    for part, off in ((first, 0), (second, 1234)):
        d_data, d_meta = gpu.upload(part.data), gpu.upload(part.meta)
        gpu.map_async(d_data, part.words, d_meta, part.count, d_out, out_offset_digests=off)
        gpu.sync()
    out = gpu.download(d_out, 32 * 3000).reshape(-1, 8)
    assert (out == oracle.leaves_packed(b.data, b.meta)).all()


# ---- reduce ----------------------------------------------------------------------

COUNTS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 128, 129, 130, 191, 193, 255, 256, 257, 383, 385, 1000, 1023,
          1151, 1153, 1919, 1921, 1985,
          1024, 1025, 2047, 2048, 2049, 2050, 4095, 4097, 5000, 8191, 8193, 65535, 65537, 100000, 262143, 524288,
          524289, 600001]


@pytest.mark.parametrize("variant", ["wave", "levels"])
def test_reduce_counts(gpu, oracle, variant):
    rng = np.random.default_rng(1)
    for n in COUNTS:
        leaves = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32)
        got = gpu.reduce_digests(leaves, levels_variant=(variant == "levels"))
        assert (got == oracle.root(leaves, threads=8)).all(), (variant, n)


@pytest.mark.parametrize("variant", ["wave", "levels"])
def test_reduce_to_capacity_height(gpu, oracle, variant):
    """A short last slice is reduced to the full slice height with self-pairing
    (reference Reductions.cpp:471, README.md:94)."""
    rng = np.random.default_rng(2)
    for n, h in [(1, 1), (1, 5), (1, 23), (2, 4), (3, 10), (5, 3), (100, 12), (129, 23), (130, 9), (257, 9), (1500, 14), (2047, 11), (2048, 11), (2048, 23), (2049, 13), (2049, 20),
                 (5000, 23), (70000, 17), (70000, 23), (300000, 20)]:
        leaves = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32)
        got = gpu.reduce_digests(leaves, height=h, levels_variant=(variant == "levels"))
        assert (got == oracle.reduce_height(leaves, h)).all(), (variant, n, h)


def test_reduce_height_zero_returns_the_node(gpu):
    """height 0 is only valid for one node and returns it unchanged (what the reference's Vulkan path
    does for a single leaf, Reductions.cpp:471-472; the front end never asks for it: SURVEY.md 8a Q1)."""
    import vk_merkle_roots_amd as vk
    leaf = np.arange(8, dtype=np.uint32)[None, :] + 7
    assert (gpu.reduce_digests(leaf, height=0) == leaf[0]).all()
    assert (gpu.reduce_digests(leaf, height=0, levels_variant=True) == leaf[0]).all()
    with pytest.raises(vk.VkmrError):
        gpu.reduce_digests(np.zeros((2, 8), np.uint32), height=0)


def test_reduce_does_not_modify_the_slice(gpu):
    rng = np.random.default_rng(3)
    leaves = rng.integers(0, 2**32, size=(10000, 8), dtype=np.uint32)
    d_in = gpu.upload(leaves)
    d_scratch = gpu.reduce_scratch(10000)
    d_root = gpu.alloc(32)
    gpu.reduce_async(d_in, 10000, 14, d_scratch, d_root)
    assert (gpu.download(d_in, 32 * 10000).reshape(-1, 8) == leaves).all()


def test_reduce_rejects_bad_height(gpu):
    import vk_merkle_roots_amd as vk
    d = gpu.alloc(32 * 8)
    with pytest.raises(vk.VkmrError):
        gpu.reduce_async(d, 8, 2, d, d)


def test_reduce_slices_batched(gpu, oracle):
    """Several slices, short last one, reduced by one batched call (every slice to capacity height)."""
    rng = np.random.default_rng(6)
    for cap_log2, nslices, last in [(8, 3, 256), (8, 3, 1), (10, 5, 700), (13, 4, 8191), (18, 3, 100000), (19, 2, 524288),
                                    (7, 9, 5), (15, 1, 20000), (8, 1500, 77), (10, 600, 1024), (9, 3000, 300), (12, 40, 1), (1, 70001, 1), (2, 40000, 3)]:
        cap = 1 << cap_log2
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
            assert (got[k] == want).all(), (cap_log2, nslices, last, k)


def test_combine_matches_cpu_rule(gpu, oracle):
    rng = np.random.default_rng(4)
    for n in [1, 2, 3, 8, 9, 100]:
        roots = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32)
        assert (gpu.combine(roots) == oracle.root(roots)).all(), n


# ---- whole path --------------------------------------------------------------------

def test_golden_roots(gpu, golden):
    import vk_merkle_roots_amd as vk
    for name, s in golden["streams"].items():
        if "stream_hex" in s:
            b = vk.pack_lines(bytes.fromhex(s["stream_hex"]))
        elif s.get("generator", "").startswith("rndm"):
            a = s["generator"].split()
            b = vk.rndm_packed(int(a[1]), int(a[2]), int(a[3]))
        else:
            alpha = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
            b = vk.pack_lines("".join("%02d%s\n" % (i, alpha) for i in range(16)).encode())
        assert (b.count, b.nbytes) == (s["items"], s["bytes"]), name
        assert vk.merkle_root_packed(gpu, b) == s["root"], name


@pytest.mark.parametrize("cap,batch", [(1 << 10, 700), (1 << 12, 5000), (1 << 16, 1 << 16), (1 << 17, 30000)])
def test_multi_slice_equals_single_tree(gpu, golden, cap, batch):
    """Config 2 stream cut into slices/batches of several sizes: always the golden root
    (SURVEY.md 8a Q6)."""
    import vk_merkle_roots_amd as vk
    b = vk.rndm_packed(42, 1 << 17, 127)
    whole = vk.merkle_root_packed(gpu, b)
    assert vk.merkle_root_packed(gpu, b, slice_capacity=cap, batch_strings=batch) == whole
    assert vk.merkle_root_packed(gpu, b, slice_capacity=cap, batch_strings=batch, levels_variant=True) == whole
    assert vk.merkle_root_packed_batched(gpu, b, slice_capacity=cap, batch_strings=batch) == whole


def test_ragged_multi_slice(gpu, oracle):
    import vk_merkle_roots_amd as vk
    for n in (1, 2, 1023, 1025, 3000, 4097, 10000):
        b = vk.rndm_packed(100 + n, n, 127)
        want = oracle.hex(oracle.root(oracle.leaves_packed(b.data, b.meta)))
        assert vk.merkle_root_packed(gpu, b) == want
        assert vk.merkle_root_packed(gpu, b, slice_capacity=1024, batch_strings=300) == want


# ---- proofs (the reference's to-do, README.md:118-120) -------------------------------------

def test_proofs_fold_to_the_root(gpu, oracle):
    """Every sibling equals the oracle's node at that level, and folding reproduces the root."""
    import vk_merkle_roots_amd as vk
    h = vk.host_lib()
    rng = np.random.default_rng(12)
    for n, height in [(1, 1), (2, 1), (3, 2), (5, 3), (8, 3), (9, 4), (100, 7), (129, 8), (1000, 10), (1000, 14), (4097, 13),
                      (70001, 17), (300000, 19)]:
        leaves = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32)
        # oracle levels
        levels = [leaves]
        for _ in range(height):
            cur = levels[-1]
            nxt = None
            if len(cur) <= 4096:
                nxt = np.stack([oracle.node(cur[2 * p], cur[2 * p + 1] if 2 * p + 1 < len(cur) else cur[2 * p])
                                for p in range((len(cur) + 1) // 2)])
            if nxt is None:
                break
            levels.append(nxt)
        want_root = oracle.reduce_height(leaves, height)
        d_in = gpu.upload(leaves)
        for index in sorted({0, n - 1, n // 2, (n * 2) // 3, max(0, n - 2)}):
            sib, root = gpu.proof(d_in, n, height, index)
            assert (root == want_root).all(), (n, height, index)
            folded = np.zeros(8, dtype=np.uint32)
            h.vkmr_host_cpu_fold_proof(leaves[index].ctypes.data, index, np.ascontiguousarray(sib).ctypes.data, height, folded.ctypes.data)
            assert (folded == want_root).all(), (n, height, index)
            if len(levels) == height + 1:      # small case: check each sibling against the oracle's level arrays
                for l in range(height):
                    p = index >> l
                    q = p ^ 1
                    if q >= len(levels[l]):
                        q = p
                    assert (sib[l] == levels[l][q]).all(), (n, height, index, l)


def test_proofs_written_during_the_reduction_equal_the_recomputed_ones(gpu, oracle):
    """vkmr_hip_reduce_proofs_async writes, while the root is computed, the same sibling arrays that
    vkmr_hip_proof_async obtains by reducing every sibling sub-tree again -- for K = 1 and K = 8 (and 16) leaves, on the same
    cases as the test above and on sizes that go through every kernel of the schedule (bulk passes, collapses, tail, levels
    above a lone node), with the same root as the plain reduction."""
    import vk_merkle_roots_amd as vk
    h = vk.host_lib()
    rng = np.random.default_rng(13)
    cases = [(1, 1), (2, 1), (3, 2), (5, 3), (8, 3), (9, 4), (100, 7), (129, 8), (1000, 10), (1000, 14), (4097, 13), (70001, 17), (300000, 19),
             (1 << 20, 20), ((1 << 20) + 77, 21), (3000001, 22), (1 << 22, 23)]
    for n, height in cases:
        leaves = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32)
        d_in = gpu.upload(leaves)
        want_root = gpu.reduce_digests(leaves, height)
        fixed = sorted({0, n - 1, n // 2, (n * 2) // 3, max(0, n - 2)})
        for K in (1, 8, 16):
            idx = (fixed + [int(x) for x in rng.integers(0, n, size=16)])[:K] if K > 1 else [fixed[-1]]
            sib, root = gpu.reduce_with_proofs(d_in, n, height, idx)
            assert (root == want_root).all(), (n, height, K)
            for q, index in enumerate(idx):
                folded = np.zeros(8, dtype=np.uint32)
                h.vkmr_host_cpu_fold_proof(leaves[index].ctypes.data, index, np.ascontiguousarray(sib[q]).ctypes.data, height, folded.ctypes.data)
                assert (folded == want_root).all(), (n, height, K, index)
        # sibling by sibling against the recomputing form, on a few leaves (it costs a reduction per leaf)
        for index in fixed[:3] if n > 100000 else fixed:
            want_sib, _ = gpu.proof(d_in, n, height, index)
            sib, _ = gpu.reduce_with_proofs(d_in, n, height, [index])
            assert (sib[0] == want_sib).all(), (n, height, index)
        d_in.free()


def test_reduce_with_proofs_rejects_bad_arguments(gpu):
    import vk_merkle_roots_amd as vk
    d = gpu.alloc(32 * 8)
    with pytest.raises(vk.VkmrError):
        gpu.reduce_with_proofs(d, 8, 3, [8])
    with pytest.raises(vk.VkmrError):
        gpu.reduce_with_proofs(d, 8, 3, list(range(8)) * 3)      # 24 > 16 proofs per reduction


def test_proof_rejects_bad_index(gpu):
    import vk_merkle_roots_amd as vk
    d = gpu.alloc(32 * 8)
    with pytest.raises(vk.VkmrError):
        gpu.proof(d, 8, 3, 8)


# ---- scratch budget (short runs must not outgrow the scratch sized for the full count) ----

def _guarded_scratch(gpu, nbytes, guard=1 << 20):
    buf = gpu.alloc(nbytes + guard)
    import vk_merkle_roots_amd as vk
    vk.check(gpu.lib.vkmr_hip_memset_async(gpu.index, gpu.stream, buf.at(nbytes), 0xA5, guard), "memset")
    gpu.sync()
    return buf, lambda: bool((gpu.download(buf, guard, dtype=np.uint8, offset=nbytes) == 0xA5).all())


def test_scratch_budget_reduce_slices_one_short_slice(gpu, oracle):
    """capacity 2^20, one slice of 2^20 - 256 nodes: its own schedule collapses one level in the first pass
    (786240 cells) where the full capacity's collapses two."""
    rng = np.random.default_rng(21)
    cap, last = 1 << 20, (1 << 20) - 256
    leaves = rng.integers(0, 2**32, size=(last, 8), dtype=np.uint32)
    d_in = gpu.upload(leaves)
    nbytes = gpu.lib.vkmr_hip_reduce_slices_scratch_bytes(cap, 1)
    d_scratch, intact = _guarded_scratch(gpu, nbytes)
    d_roots = gpu.alloc(32)
    gpu.reduce_slices_async(d_in, 1, cap, last, 20, d_scratch, d_roots)
    got = gpu.download(d_roots, 32)
    assert intact()
    assert (got == oracle.reduce_height(leaves, 20)).all()


def test_scratch_budget_proof_sibling_subtrees(gpu, oracle):
    """count = 3 * 2^20 - 256, index 0: the level-21 sibling is a sub-tree of 2^20 - 256 leaves."""
    import vk_merkle_roots_amd as vk
    rng = np.random.default_rng(22)
    n, height = 3 * (1 << 20) - 256, 22
    leaves = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32)
    d_in = gpu.upload(leaves)
    d_scratch, intact = _guarded_scratch(gpu, gpu.lib.vkmr_hip_reduce_scratch_bytes(n))
    d_sib, d_root = gpu.alloc(32 * height), gpu.alloc(32)
    want_root = oracle.reduce_height(leaves, height)
    h = vk.host_lib()
    for index in (0, n - 1, (1 << 21) + 5):
        vk.check(gpu.lib.vkmr_hip_proof_async(gpu.index, gpu.stream, d_in.ptr, n, height, index, d_scratch.ptr, d_sib.ptr, d_root.ptr), "proof")
        sib = gpu.download(d_sib, 32 * height).reshape(-1, 8)
        assert intact(), index
        assert (gpu.download(d_root, 32) == want_root).all()
        folded = np.zeros(8, dtype=np.uint32)
        h.vkmr_host_cpu_fold_proof(leaves[index].ctypes.data, index, np.ascontiguousarray(sib).ctypes.data, height, folded.ctypes.data)
        assert (folded == want_root).all(), index
