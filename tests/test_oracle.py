"""The oracle (oracle/sha256d_oracle.c) pinned against the reference: golden vectors
recorded from the reference's own CPU path (tests/golden/vectors.json), the compiled
reference itself (oracle/_ref, when present) and an independent hashlib model."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import REF_DIR, golden_pattern


def d256(b):
    return hashlib.sha256(hashlib.sha256(b).digest()).digest()


def model_root(strings):
    level = [d256(s) for s in strings]
    while True:
        if len(level) % 2:
            level.append(level[-1])
        level = [d256(level[i] + level[i + 1]) for i in range(0, len(level), 2)]
        if len(level) == 1:
            return level[0].hex()


def test_leaf_golden(oracle, golden):
    for v in golden["leaves"]:
        msg = golden_pattern(v["len"], v["salt"])
        assert oracle.sha256(msg).hex() == v["sha256"], v["len"]
        assert oracle.hex(oracle.leaf(msg)) == v["sha256d"], v["len"]


def test_leaf_every_length_vs_hashlib(oracle):
    for n in list(range(0, 300)) + [511, 512, 513, 4095, 4096, 4097, 10000]:
        msg = golden_pattern(n, 77 + n)
        assert oracle.hex(oracle.leaf(msg)) == d256(msg).hex(), n


def test_tree_golden(oracle, golden):
    for t in golden["trees"]:
        n = t["count"]
        strs = [golden_pattern(1 + (i * 7) % 40, 1000 + i) for i in range(n)]
        leaves = np.stack([oracle.leaf(s) for s in strs])
        assert oracle.hex(oracle.root(leaves)) == t["root"], n
        assert model_root(strs) == t["root"], n
        if n > 1:
            assert oracle.hex(oracle.root(leaves, threads=3)) == t["root"], n


def test_single_leaf_is_hashed_with_itself(oracle):
    # CpuSha256D::Root's do-while runs once for one leaf
    leaf = oracle.leaf(b"x")
    assert (oracle.root(leaf[None, :]) == oracle.node(leaf, leaf)).all()


def test_reduce_height_matches_global_tree(oracle):
    # full slices + a short last slice reduced to capacity height +
    # duplicate-last combine == one global duplicate-last tree
    rng = np.random.default_rng(5)
    for cap_log2, n in [(2, 11), (3, 8), (3, 9), (3, 17), (4, 50), (2, 4), (5, 33), (5, 32), (3, 24)]:
        cap = 1 << cap_log2
        leaves = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32)
        roots = []
        for lo in range(0, n, cap):
            sl = leaves[lo: lo + cap]
            nslices = (n + cap - 1) // cap
            h = cap_log2 if nslices > 1 else max(1, int(sl.shape[0] - 1).bit_length())
            roots.append(oracle.reduce_height(sl, h))
        top = roots[0] if len(roots) == 1 else oracle.root(np.stack(roots))
        assert (top == oracle.root(leaves)).all(), (cap, n)


def test_streams_golden(oracle, golden):
    for name, s in golden["streams"].items():
        if "stream_hex" not in s:
            continue
        root, cnt, nb = oracle.root_of_stream(bytes.fromhex(s["stream_hex"]))
        assert (root, cnt, nb) == (s["root"], s["items"], s["bytes"]), name


def test_empty_stream_has_no_root(oracle):
    assert oracle.root_of_stream(b"") == ("", 0, 0)
    assert oracle.root_of_stream(b"\n\n\n") == ("", 0, 0)


def test_rndm_streams_golden(oracle, golden, native):
    """rndm-generated golden streams: regenerate with OUR rndm (bit-identical stream,
    checked by sha256), run the oracle, compare with the reference's recorded root."""
    rndm = os.path.join(os.path.dirname(native.HIP_LIB), "bin", "rndm")
    for name, s in golden["streams"].items():
        if not s.get("generator", "").startswith("rndm"):
            continue
        args = s["generator"].split()[1:]
        if int(args[1]) > 200000:
            continue   # G3 (2^20 leaves) is checked on the GPU path and in test_g3_stream below
        data = subprocess.run([rndm] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        assert hashlib.sha256(data).hexdigest() == s["stream_sha256"], name
        assert oracle.root_of_stream(data) == (s["root"], s["items"], s["bytes"]), name


def test_g1_strm(oracle, golden, native):
    strm = os.path.join(os.path.dirname(native.HIP_LIB), "bin", "strm")
    alpha = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
    data = subprocess.run([strm] + ["%02d%s" % (i, alpha) for i in range(16)], stdout=subprocess.PIPE).stdout
    g = golden["streams"]["G1_strm16x64"]
    assert hashlib.sha256(data).hexdigest() == g["stream_sha256"]
    assert oracle.root_of_stream(data) == (g["root"], g["items"], g["bytes"])


def test_g3_stream_packed_mt(oracle, golden):
    """Config 2 (rndm 42 2^20 127) through the packed + threaded oracle entry points."""
    import vk_merkle_roots_amd as vk
    g = golden["streams"]["G3_rndm_42_1048576_127"]
    b = vk.rndm_packed(42, 1 << 20, 127)
    assert (b.count, b.nbytes) == (g["items"], g["bytes"])
    leaves = oracle.leaves_packed(b.data, b.meta, threads=8)
    assert oracle.hex(oracle.root(leaves, threads=8)) == g["root"]


def test_oracle_vs_compiled_reference(oracle, ref_lib):
    """Leaf by leaf and root by root against the reference's own code (oracle/_ref)."""
    rng = np.random.default_rng(11)
    for n in list(range(0, 200, 3)) + [1000, 5000]:
        msg = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        out = C.create_string_buffer(32)
        ref_lib.ref_sha256d(msg, n, out)
        assert oracle.hex(oracle.leaf(msg)) == out.raw.hex(), n
    for n in [1, 2, 3, 5, 8, 13, 64, 77, 256, 301]:
        strs = [rng.integers(0, 256, size=int(rng.integers(1, 200)), dtype=np.uint8).tobytes() for _ in range(n)]
        blob = b"".join(strs)
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(s) for s in strs])
        hexbuf = C.create_string_buffer(65)
        ref_lib.ref_root(blob, offs.ctypes.data, n, hexbuf)
        leaves = np.stack([oracle.leaf(s) for s in strs])
        assert oracle.hex(oracle.root(leaves)) == hexbuf.value.decode(), n


def test_reference_driver_matches_golden(golden):
    exe = os.path.join(REF_DIR, "vkmr_cpu_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref not built")
    s = golden["streams"]["L2_empty_lines"]
    out = subprocess.run([exe], input=bytes.fromhex(s["stream_hex"]), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    assert f"CPU: computed root (of {s['items']} item(s), {s['bytes']} byte(s)) => {s['root']} in " in out


def test_big_roots_fixture_is_consistent(oracle):
    """tests/golden/big_roots.json (the reference CPU path on rndm 42..49 2^26 127): combined[N] is the duplicate-last
    tree over the first N sub-roots in order (reference Reductions.cpp:703-712), N = 1 the root itself."""
    import json
    import os
    from conftest import ROOT
    big = json.load(open(os.path.join(ROOT, "tests", "golden", "big_roots.json")))
    assert big["count"] == 1 << 26 and big["maxlen"] == 127
    subs = [big["sub_roots"][str(42 + r)] for r in range(8)]
    assert all(s["items"] == 1 << 26 for s in subs) and len({s["root"] for s in subs}) == 8
    words = np.array([np.frombuffer(bytes.fromhex(s["root"]), dtype=">u4").astype(np.uint32) for s in subs], dtype=np.uint32)
    for n in range(1, 9):
        want = subs[0]["root"] if n == 1 else oracle.hex(oracle.root(words[:n]))
        assert big["combined"][str(n)] == want, n
