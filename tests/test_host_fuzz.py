"""Host-side robustness: random streams through the front end's line rules and CPU backend versus the
oracle (hypothesis), and the same front end built with AddressSanitizer + UBSan (CPU build only)."""
import os
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from conftest import ROOT

ALPHABET = [b"a", b"b", b"\r", b"\n", b"\n", b"\xff", b"\x00", b" "]


@settings(max_examples=120, deadline=None)
@given(st.lists(st.sampled_from(ALPHABET), max_size=200), st.integers(0, 3))
def test_pack_lines_and_cpu_backend_match_oracle(native, oracle, parts, reps):
    import vk_merkle_roots_amd as vk
    stream = b"".join(parts) * (reps + 1)
    want_root, want_count, want_bytes = oracle.root_of_stream(stream)
    b = vk.pack_lines(stream)
    assert (b.count, b.nbytes) == (want_count, want_bytes)
    if b.count == 0:
        return
    h = vk.host_lib()
    leaves = np.zeros((b.count, 8), dtype=np.uint32)
    h.vkmr_host_cpu_leaves(b.data.ctypes.data, b.meta.ctypes.data, b.count, leaves.ctypes.data)
    top = np.zeros(8, dtype=np.uint32)
    assert h.vkmr_host_cpu_combine(leaves.ctypes.data, b.count, top.ctypes.data) == 0
    assert vk.engine.digest_hex(top) == want_root


@settings(max_examples=25, deadline=None)
@given(st.lists(st.binary(min_size=0, max_size=300).filter(lambda s: b"\n" not in s), min_size=0, max_size=40), st.booleans())
def test_vkmr_cpu_binary_matches_oracle(native, oracle, lines, trailing_newline):
    stream = b"\n".join(lines) + (b"\n" if trailing_newline else b"")
    want_root, want_count, want_bytes = oracle.root_of_stream(stream)
    exe = os.path.join(os.path.dirname(native.HIP_LIB), "bin", "vkmr")
    r = subprocess.run([exe, "CPU"], input=stream, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0
    out = [l for l in r.stdout.decode().splitlines() if "computed root" in l]
    if want_count == 0:
        assert not out
    else:
        assert f"(of {want_count} item(s), {want_bytes} byte(s)) => {want_root} in " in out[0]


def test_front_end_under_asan_ubsan(native, oracle, tmp_path):
    """The whole host front end ("CPU" backend) compiled with -fsanitize=address,undefined."""
    host = os.path.join(ROOT, "vk_merkle_roots_amd", "csrc", "host")
    libdir = os.path.dirname(native.HIP_LIB)
    exe = str(tmp_path / "vkmr_asan")
    srcs = [os.path.join(host, f) for f in ("vkmr_main.cpp", "cpu_sha256d.cpp", "hip_sha256d.cpp", "inputs.cpp", "batches.cpp",
                                            "slices.cpp", "mappings.cpp", "reductions.cpp", "stream_pack.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "include"), "-I", host, "-o", exe] + srcs +
                          ["-L", libdir, "-lvkmr_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    rng = np.random.default_rng(3)
    streams = [b"", b"\n", b"a", b"a\nb\r\n\n\nccc", b"x" * 3000000 + b"\n" + b"y" * 10 + b"\n" + b"z" * 2500000,
               b"\n".join(rng.integers(32, 126, size=int(n), dtype=np.uint8).tobytes() for n in rng.integers(0, 400, size=3000))]
    for s in streams:
        r = subprocess.run([exe, "CPU"], input=s, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        want_root, want_count, _ = oracle.root_of_stream(s)
        out = [l for l in r.stdout.decode().splitlines() if "computed root" in l]
        assert (not out) if want_count == 0 else (want_root in out[0])


def test_fork_join_pool_under_tsan(tmp_path):
    """The packer's thread pool and the two-pass parallel packing scheme under ThreadSanitizer."""
    host = os.path.join(ROOT, "vk_merkle_roots_amd", "csrc", "host")
    exe = str(tmp_path / "fj_tsan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-I", host, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "fork_join_test.cpp"), os.path.join(host, "stream_pack.cpp"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and r.stdout.startswith(b"ok "), (r.stdout, r.stderr[-1500:])


@settings(max_examples=300, deadline=None)
@given(st.lists(st.one_of(st.sampled_from([b"\n", b"\n\n", b"\r\n", b"x" * 63 + b"\n", b"y" * 64 + b"\n", b"z" * 65, b"q" * 200, b"\xff\x00"]),
                          st.binary(min_size=0, max_size=90)), max_size=60),
       st.integers(0, 5000), st.integers(0, 400), st.booleans(), st.integers(0, 7))
def test_avx2_line_splitter_equals_the_portable_one(native, parts, cap_words, cap_meta, final, first_word):
    """PackLines / CountLines find newlines 64 bytes at a time (AVX2) where the CPU can; line for line they must do what the
    memchr-per-line forms do: same strings, same packed words, same bytes consumed when a buffer fills up mid-stream, same
    treatment of an unfinished last line (final or not)."""
    import ctypes as C
    import vk_merkle_roots_amd as vk
    h = vk.host_lib()
    stream = b"".join(parts)
    buf = np.frombuffer(stream, dtype=np.uint8) if stream else np.zeros(0, np.uint8)
    ptr = buf.ctypes.data if len(stream) else None
    a, b = np.zeros(5, np.uint64), np.zeros(5, np.uint64)
    h.vkmr_host_count_lines(ptr, len(stream), 0, a.ctypes.data)
    h.vkmr_host_count_lines(ptr, len(stream), 1, b.ctypes.data)
    assert a.tolist() == b.tolist()
    outs = []
    for which in (0, 1):
        data = np.full(first_word + cap_words + 8, 0xABABABAB, dtype=np.uint32)
        meta = np.zeros((cap_meta + 1, 2), dtype=np.uint32)
        out = np.zeros(5, np.uint64)
        h.vkmr_host_pack_prefix(ptr, len(stream), int(final), data.ctypes.data, first_word, first_word + cap_words, meta.ctypes.data, cap_meta, which,
                                out.ctypes.data)
        outs.append((out.tolist(), data.tolist(), meta.tolist()))
    assert outs[0] == outs[1]
    assert outs[0][1][first_word + cap_words:] == [0xABABABAB] * 8      # nothing written past the capacity


@settings(max_examples=300, deadline=None)
@given(st.lists(st.one_of(st.sampled_from([b"\n", b"\n\n", b"\r\n", b"x" * 63 + b"\n", b"y" * 64 + b"\n", b"z" * 65, b"q" * 200, b"\xff\x00",
                                           b"p" * 127 + b"\n", b"r" * 128 + b"\n", b"s" * 129 + b"\n"]),
                          st.binary(min_size=0, max_size=140)), max_size=60),
       st.integers(0, 7))
def test_indexed_two_pass_packer_equals_the_portable_one(native, parts, first_word):
    """What the parallel packer runs on each part of a span -- pass 1 records where the lines end (four positions per 64-byte
    block written whatever the count), pass 2 moves lines of up to 128 bytes as four vectors and clears the bytes behind
    their ends -- must place the same words and metadata as the memchr-and-memcpy form, write nothing outside the words
    pass 1 announced (the next words are another thread's part), and leave no stray byte in a string's last word."""
    import vk_merkle_roots_amd as vk
    h = vk.host_lib()
    stream = b"".join(parts)
    buf = np.frombuffer(stream, dtype=np.uint8) if stream else np.zeros(0, np.uint8)
    ptr = buf.ctypes.data if len(stream) else None
    cap_meta = stream.count(b"\n") + 1
    cap_words = len(stream) // 4 + cap_meta + 1
    ref_data = np.full(first_word + cap_words + 8, 0xABABABAB, dtype=np.uint32)
    ref_meta = np.zeros((cap_meta + 1, 2), dtype=np.uint32)
    ref = np.zeros(5, np.uint64)
    h.vkmr_host_pack_prefix(ptr, len(stream), 1, ref_data.ctypes.data, first_word, first_word + cap_words, ref_meta.ctypes.data, cap_meta, 1, ref.ctypes.data)
    consumed, strings, words, nbytes, empties = (int(v) for v in ref)
    assert consumed == len(stream)
    for which in (0, 1, 2):      # the vector form, the portable one, the vector form with streaming stores
        data = np.full(first_word + cap_words + 8, 0xABABABAB, dtype=np.uint32)
        meta = np.zeros((cap_meta + 1, 2), dtype=np.uint32)
        out = np.zeros(4, np.uint64)
        got = h.vkmr_host_pack_indexed(ptr, len(stream), data.ctypes.data, first_word, first_word + cap_words, meta.ctypes.data, cap_meta, which, out.ctypes.data)
        assert got == strings and [int(v) for v in out[:3]] == [words, nbytes, empties]
        assert data.tolist() == ref_data.tolist()          # incl. the untouched words before first_word and from first_word + words on
        assert meta.tolist() == ref_meta.tolist()


@settings(max_examples=60, deadline=None)
@given(st.lists(st.one_of(st.sampled_from([b"\n", b"\n\n\n", b"x" * 63 + b"\n", b"y" * 64, b"\n" * 70, b"z" * 130 + b"\n\n"]), st.binary(min_size=0, max_size=100)), max_size=40),
       st.booleans())
def test_copy_and_count_lines(native, parts, after_newline):
    """The device-side splitter's host pass: the bytes copied as they are, newlines counted, and those that end an empty
    line (the byte before is a newline, or the stream's start) -- both forms against a plain Python count."""
    import vk_merkle_roots_amd as vk
    h = vk.host_lib()
    stream = b"".join(parts)
    buf = np.frombuffer(stream, dtype=np.uint8) if stream else np.zeros(0, np.uint8)
    want_nl = stream.count(b"\n")
    prev, want_empty = after_newline, 0
    for ch in stream:
        want_empty += ch == 10 and prev
        prev = ch == 10
    for which in (0, 1):
        dst = np.full(len(stream) + 8, 0xEE, dtype=np.uint8)
        out = np.zeros(2, np.uint64)
        h.vkmr_host_copy_and_count(buf.ctypes.data if len(stream) else None, len(stream), dst.ctypes.data, int(after_newline), which, out.ctypes.data)
        assert [int(v) for v in out] == [want_nl, want_empty]
        assert dst[:len(stream)].tobytes() == stream and (dst[len(stream):] == 0xEE).all()


def test_streaming_store_form_of_the_packer_on_large_parts(native):
    """PackIndexed with streaming stores assembles lines in an 8 KiB window and sends whole 64-byte lines out: parts much
    larger than the window, lines longer than its 128-byte fast path, destinations at every word offset within a line
    (the part's first and last lines are shared with other parts) -- same words, same entries, nothing outside."""
    import vk_merkle_roots_amd as vk
    h = vk.host_lib()
    rng = np.random.default_rng(5)
    for hi, count in ((127, 30000), (300, 8000), (3, 50000), (5000, 300)):
        stream = b"".join(rng.integers(33, 126, size=int(k), dtype=np.uint8).tobytes() + b"\n" for k in rng.integers(0, hi + 1, size=count))
        buf = np.frombuffer(stream, dtype=np.uint8)
        cap_meta = stream.count(b"\n") + 1
        cap_words = len(stream) // 4 + cap_meta + 1
        for first_word in (0, 1, 7, 15, 16, 33):
            outs = []
            for which in (1, 2):
                data = np.full(first_word + cap_words + 40, 0xABABABAB, dtype=np.uint32)
                meta = np.zeros((cap_meta + 1, 2), dtype=np.uint32)
                out = np.zeros(4, np.uint64)
                got = h.vkmr_host_pack_indexed(buf.ctypes.data, len(stream), data.ctypes.data, first_word, first_word + cap_words, meta.ctypes.data, cap_meta, which,
                                               out.ctypes.data)
                outs.append((got, out.tolist(), data, meta))
            assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
            assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3]), (hi, first_word)
