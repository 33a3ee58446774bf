"""Host-side helpers: the restated glibc rand(), the rndm/strm tools, line packing."""
import ctypes as C
import ctypes.util
import hashlib
import os
import subprocess

import numpy as np
import pytest


def test_glibc_rand_restatement_matches_libc(native):
    import vk_merkle_roots_amd as vk
    libc = C.CDLL(ctypes.util.find_library("c"))
    libc.rand.restype = C.c_int
    h = vk.host_lib()
    for seed in (1, 42, 1712489279, 0, 0xFFFFFFFF):
        libc.srand(C.c_uint(seed))
        want = np.array([libc.rand() for _ in range(5000)], dtype=np.int32)
        got = np.zeros(5000, dtype=np.int32)
        h.vkmr_host_rndm_rand(seed, got.ctypes.data, 5000)
        assert (want == got).all(), seed


def test_rndm_tool_stream_matches_reference_tool(native, golden):
    rndm = os.path.join(os.path.dirname(native.HIP_LIB), "bin", "rndm")
    g = golden["streams"]["G2_rndm_1712489279_1024_127"]
    r = subprocess.run([rndm, "1712489279", "1024", "127"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert hashlib.sha256(r.stdout).hexdigest() == g["stream_sha256"]
    assert b"Using seed: 1712489279" in r.stderr and b"Wrote 1024 string(s) in a total of 63646 byte(s)." in r.stderr


def test_rndm_packed_equals_packing_the_stream(native):
    import vk_merkle_roots_amd as vk
    rndm = os.path.join(os.path.dirname(native.HIP_LIB), "bin", "rndm")
    data = subprocess.run([rndm, "9", "500", "300"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    a = vk.pack_lines(data)
    b = vk.rndm_packed(9, 500, 300)
    assert a.count == b.count == 500 and a.words == b.words and a.nbytes == b.nbytes
    assert (a.meta == b.meta).all() and (a.data == b.data).all()


def test_pack_lines_rules(native):
    import vk_merkle_roots_amd as vk
    b = vk.pack_lines(b"\n\none\r\n\ntwo\n\n\nthree")
    assert b.count == 3
    assert b.meta.tolist() == [[0, 4], [1, 3], [2, 5]]
    raw = b.data.tobytes()
    assert raw[0:4] == b"one\r" and raw[4:7] == b"two" and raw[7] == 0 and raw[8:13] == b"three"
    assert vk.pack_lines(b"").count == 0
    assert vk.pack_lines(b"\n").count == 0
    assert vk.pack_lines(b"x").meta.tolist() == [[0, 1]]


def test_batch_slice_rebases_metadata(native):
    import vk_merkle_roots_amd as vk
    b = vk.rndm_packed(3, 100, 50)
    s = b.slice(10, 20)
    assert s.count == 10 and s.meta[0, 0] == 0
    for i in range(10):
        st, sz = s.meta[i]
        st0, sz0 = b.meta[10 + i]
        assert sz == sz0
        assert s.data[st: st + (sz + 3) // 4].tolist() == b.data[st0: st0 + (sz0 + 3) // 4].tolist()


def test_rndm_stream_batches_equal_one_shot(native):
    import vk_merkle_roots_amd as vk
    whole = vk.rndm_packed(13, 3000, 300)
    st = vk.RndmStream(13, 300)
    parts = [st.next(1000), st.next(1500), st.next(500)]
    st.close()
    lo = 0
    for p in parts:
        ref = whole.slice(lo, lo + p.count)
        assert (p.meta == ref.meta).all() and (p.data == ref.data).all()
        lo += p.count


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` must never report a smaller N: with fewer than N GPUs (none here, one on the GPU box
    for N = 2) it exits non-zero before starting any rank and prints no result line."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "9", "--steps", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert r.returncode != 0
    assert r.stdout.strip() == b""
    assert b"refusing to run fewer ranks" in r.stderr
    # under a launcher the rank count must be the requested one too
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300, env=env)
    assert r.returncode != 0 and r.stdout.strip() == b""


def test_pack_tuner_picks_the_faster_form(tmp_path):
    """PackTuner: alternates at first, then uses the form of the packer's second pass (ordinary or streaming stores) that has
    moved more bytes per second of late, probing the other every sixteenth call; follows a change of the host's state;
    can be forced (tests/c/pack_tuner_test.cpp)."""
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(ROOT, "vk_merkle_roots_amd", "csrc", "host")
    exe = str(tmp_path / "pack_tuner_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", host, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "pack_tuner_test.cpp"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout.strip() == b"ok", (r.stdout, r.stderr)
