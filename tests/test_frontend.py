"""The C++ front end (`vkmr`): same command line and output line as the reference's
program, "CPU" backend everywhere, "hip:<n>" on the GPU box."""
import hashlib
import os
import re
import subprocess

import pytest

LINE = re.compile(r"^(?P<name>\S+): computed root \(of (?P<items>\d+) item\(s\), (?P<bytes>\d+) byte\(s\)\) => (?P<root>[0-9a-f]{64}) in [0-9.e+-]+$")


def virtual_devices_env(k):
    """Environment that makes the box's one GPU look like k GPUs to the unmodified product binaries:
    the LD_PRELOAD test double tests/c/virt_devices.cpp (device enumeration + a same-process stand-in for
    RCCL, which itself refuses several ranks on one physical GPU)."""
    from conftest import build_virt_devices
    return {"LD_PRELOAD": build_virt_devices(), "VKMR_TEST_VIRTUAL_DEVICES": str(k)}


def tool(native, name):
    return os.path.join(os.path.dirname(native.HIP_LIB), "bin", name)


def run_vkmr(native, backend, stream, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([tool(native, "vkmr")] + ([backend] if backend else []), input=stream, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=e, timeout=600)
    out = r.stdout.decode().splitlines()
    res = [LINE.match(l) for l in out]
    res = [m for m in res if m]
    return r, out, (res[-1].groupdict() if res else None)


def golden_stream(native, s):
    if "stream_hex" in s:
        return bytes.fromhex(s["stream_hex"])
    if s.get("generator", "").startswith("rndm"):
        return subprocess.run([tool(native, "rndm")] + s["generator"].split()[1:], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    alpha = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
    return subprocess.run([tool(native, "strm")] + ["%02d%s" % (i, alpha) for i in range(16)], stdout=subprocess.PIPE).stdout


def test_cpu_backend_golden_streams(native, golden):
    """BASELINE configs[0] (16 fixed 64 B strings via strm | vkmr CPU) and every other golden stream."""
    for name, s in golden["streams"].items():
        if name.startswith("G3"):
            continue
        r, out, m = run_vkmr(native, "CPU", golden_stream(native, s))
        assert r.returncode == 0 and out[0] == "Initializing for: CPU", name
        assert m and (m["name"], int(m["items"]), int(m["bytes"]), m["root"]) == ("CPU", s["items"], s["bytes"], s["root"]), name


def test_cpu_backend_config2(native, golden):
    s = golden["streams"]["G3_rndm_42_1048576_127"]
    data = golden_stream(native, s)
    assert hashlib.sha256(data).hexdigest() == s["stream_sha256"]
    r, out, m = run_vkmr(native, "CPU", data)
    assert m and (int(m["items"]), int(m["bytes"]), m["root"]) == (s["items"], s["bytes"], s["root"])


def test_empty_input_prints_no_root(native):
    r, out, m = run_vkmr(native, "CPU", b"")
    assert r.returncode == 0 and m is None and out == ["Initializing for: CPU"]
    assert b"Read an empty string?" in r.stderr
    r, out, m = run_vkmr(native, "CPU", b"\n\n")
    assert m is None and r.stderr.count(b"Read an empty string?") == 3


def test_a_read_error_is_not_the_end_of_the_stream(native):
    """a read(2) that FAILS (EISDIR here: stdin is a directory; EIO on a dying disk) used to be taken for the end
    of the stream -- a root over whatever had arrived, exit code 0.  Now: the error on stderr, no root, exit code 2."""
    fd = os.open("/tmp", os.O_RDONLY)
    try:
        r = subprocess.run([tool(native, "vkmr"), "CPU"], stdin=fd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    finally:
        os.close(fd)
    assert r.returncode == 2 and b"computed root" not in r.stdout and b"Reading the input failed" in r.stderr, (r.returncode, r.stderr)
    # an ordinary stream right behind it is unaffected
    r, out, m = run_vkmr(native, "CPU", b"a\nb\n")
    assert r.returncode == 0 and m and m["items"] == "2"


def test_unknown_backend_aborts(native):
    r, out, m = run_vkmr(native, "no-such-device", b"a\n")
    assert r.returncode == 1 and b"No device selected; aborting." in r.stderr


def test_default_backend_selection(native):
    """No argument: the only backend is picked; with several the program lists them and exits 1
    (reference Vkmr.cpp:72-84)."""
    r, out, m = run_vkmr(native, None, b"a\nb\n")
    if r.returncode == 0:
        assert out[0] == "Initializing for: CPU" and m["root"]
    else:
        assert r.returncode == 1 and b"Available:" in r.stderr and b"* CPU" in r.stderr and b"* hip:0" in r.stderr


def test_input_reader_block_boundaries(native, oracle):
    """Lines straddling the reader's 1 MiB blocks, a very long line, CR kept."""
    lines = [b"x" * 700000, b"y" * 700000, b"z" * 3000000, b"tail\r"]
    data = b"\n".join(lines) + b"\n"
    r, out, m = run_vkmr(native, "CPU", data)
    want, cnt, nb = oracle.root_of_stream(data)
    assert m and (m["root"], int(m["items"]), int(m["bytes"])) == (want, cnt, nb)


@pytest.mark.gpu
def test_hip_backend_golden_streams(native, golden):
    for name, s in golden["streams"].items():
        r, out, m = run_vkmr(native, "hip:0", golden_stream(native, s))
        assert r.returncode == 0 and out[0] == "Initializing for: hip:0", (name, r.stderr[-500:])
        assert m and (m["name"], int(m["items"]), int(m["bytes"]), m["root"]) == ("hip:0", s["items"], s["bytes"], s["root"]), name


@pytest.mark.gpu
@pytest.mark.parametrize("env", [
    {"VKMR_SLICE_LOG2": "10", "VKMR_BATCH_BYTES": "65536", "VKMR_MAX_INFLIGHT": "2"},
    {"VKMR_SLICE_LOG2": "12", "VKMR_BATCH_BYTES": "20000", "VKMR_MAX_INFLIGHT": "1"},
    {"VKMR_SLICE_LOG2": "16", "VKMR_BATCH_MB": "1", "VKMR_VERBOSE": "1"},
    {"VKMR_SLICE_LOG2": "20", "VKMR_BATCH_MB": "4"},
    {"VKMR_SEND_METADATA": "1"},                                   # batches cross as data + 8-byte entries, as the reference sends them
    {"VKMR_SEND_METADATA": "1", "VKMR_SLICE_LOG2": "11", "VKMR_BATCH_BYTES": "50000"},
])
def test_hip_backend_small_slices_and_batches(native, golden, env):
    """Many batches per slice, many slices, back-pressure: always the golden root."""
    for name in ("G2_rndm_1712489279_1024_127", "G6_rndm_7_1000_300", "G3_rndm_42_1048576_127", "L3_no_trailing_newline"):
        s = golden["streams"][name]
        r, out, m = run_vkmr(native, "hip:0", golden_stream(native, s), env)
        assert r.returncode == 0, (name, r.stderr[-500:])
        assert m and (int(m["items"]), int(m["bytes"]), m["root"]) == (s["items"], s["bytes"], s["root"]), (name, env)


@pytest.mark.gpu
def test_hip_backend_long_strings(native, golden):
    s = golden["streams"]["G4_rndm_42_4096_4096"]
    r, out, m = run_vkmr(native, "hip:0", golden_stream(native, s), {"VKMR_SLICE_LOG2": "11", "VKMR_BATCH_BYTES": "1000000"})
    assert m and m["root"] == s["root"]


@pytest.mark.gpu
def test_hip_backend_empty_input(native):
    r, out, m = run_vkmr(native, "hip:0", b"\n")
    assert r.returncode == 0 and m is None


@pytest.mark.gpu
def test_hip_string_larger_than_batch_is_refused(native):
    r, out, m = run_vkmr(native, "hip:0", b"a\n" + b"b" * 10000 + b"\nc\n", {"VKMR_BATCH_BYTES": "4096", "VKMR_BATCH_MAX_MB": "0"})
    assert b"does not fit an empty batch" in r.stderr
    assert m and int(m["items"]) == 1   # the loop stops at the refused string, like the reference (Vkmr.cpp:44-47)


@pytest.mark.gpu
def test_hip_string_larger_than_batch_grows_the_batches(native, oracle):
    """Batches start at 64 MiB here but 256 MiB in the reference: a string the reference would take is not refused,
    the batches grow up to VKMR_BATCH_MAX_MB instead."""
    data = b"a\n" + b"b" * 10000 + b"\nc\n" + b"d" * 70000 + b"\ne\n"
    want, cnt, nb = oracle.root_of_stream(data)
    r, out, m = run_vkmr(native, "hip:0", data, {"VKMR_BATCH_BYTES": "4096"})
    assert m and (m["root"], int(m["items"]), int(m["bytes"])) == (want, cnt, nb), r.stderr[-300:]


@pytest.mark.gpu
def test_hip_slices_and_scratch_are_recycled(native, golden):
    """README.md:113 (the reference's first to-do): 256 slices of 2^12 stream through at most max_inflight + 1
    slice allocations; the memory of a retired slice, its reduction's scratch and the batches are re-used."""
    s = golden["streams"]["G3_rndm_42_1048576_127"]
    r, out, m = run_vkmr(native, "hip:0", golden_stream(native, s), {"VKMR_SLICE_LOG2": "12", "VKMR_VERBOSE": "1", "VKMR_MAX_INFLIGHT": "4"})
    assert m and m["root"] == s["root"]
    assert sum(1 for l in out if l.startswith("Looking for")) <= 5
    summary = [l for l in out if l.startswith("Allocations:")][0]
    slices, batches, scratch = [int(x) for x in re.findall(r"(\d+) (?:slice\(s\),|batch\(es\),|reduction)", summary)]
    assert slices <= 5 and batches <= 6 and scratch <= 5, summary
    assert "for 256 slice(s)" in summary


@pytest.mark.gpu
@pytest.mark.parametrize("budget", ["1", "2"])
def test_hip_capped_slice_budget_waits_instead_of_halting(native, golden, budget):
    """HBM "capped" to one or two slices: Add() blocks on the oldest reduction and takes over its slice
    (the reference halts: SHA-256vk.cpp:396-399); the root is still the golden one."""
    for name, log2 in (("G3_rndm_42_1048576_127", "14"), ("G6_rndm_7_1000_300", "6")):
        s = golden["streams"][name]
        r, out, m = run_vkmr(native, "hip:0", golden_stream(native, s),
                             {"VKMR_SLICE_LOG2": log2, "VKMR_SLICE_BUDGET": budget, "VKMR_BATCH_BYTES": "200000"})
        assert r.returncode == 0, r.stderr[-400:]
        assert m and (int(m["items"]), m["root"]) == (s["items"], s["root"]), (name, budget)
        assert sum(1 for l in out if l.startswith("Looking for")) <= int(budget)


@pytest.mark.gpu
def test_hip_long_strings_grow_the_batches(native, golden):
    """Strings of 2 KiB on average: the front end moves to larger batches (a launch needs ~2^19 strings to fill
    the chip) -- same root."""
    s = golden["streams"]["G4_rndm_42_4096_4096"]
    r, out, m = run_vkmr(native, "hip:0", golden_stream(native, s), {"VKMR_BATCH_BYTES": "1000000", "VKMR_BATCH_MAX_MB": "256", "VKMR_VERBOSE": "1"})
    assert m and m["root"] == s["root"]
    assert any("batches of" in l for l in out)


@pytest.mark.gpu
def test_hip_slice_size_beyond_hbm_is_clamped_not_fatal(native, golden):
    """A slice larger than HBM is not discovered by a failed hipMalloc any more: the capacity is derived from the device's free
    memory (reference Slices<T>::SliceSize, src/vkmr/Slices.h:421-454) and a VKMR_SLICE_LOG2 beyond it is clamped with a
    message; the root is the golden one.  (The failure path itself -- Add() refuses, nothing is printed, exit code 0,
    src/vkmr/Vkmr.cpp:44-52 -- is exercised with real allocation failures on CPU: tests/test_host_pipeline.py.)"""
    s = golden["streams"]["G2_rndm_1712489279_1024_127"]
    r, out, m = run_vkmr(native, "hip:0", golden_stream(native, s), {"VKMR_SLICE_LOG2": "36", "VKMR_VERBOSE": "1"})
    assert r.returncode == 0 and m and m["root"] == s["root"], r.stderr[-400:]
    assert b"VKMR_SLICE_LOG2=36 does not fit the device memory" in r.stderr
    line = [l for l in out if l.startswith("Slices of 2^")]
    assert line and 20 <= int(line[0].split("2^")[1].split()[0]) < 36, line


@pytest.mark.gpu
@pytest.mark.parametrize("ndev,env", [
    (2, {"VKMR_SLICE_LOG2": "12", "VKMR_BATCH_BYTES": "50000"}),
    (4, {"VKMR_SLICE_LOG2": "16", "VKMR_BATCH_MB": "1", "VKMR_MAX_INFLIGHT": "3"}),
    (3, {"VKMR_SLICE_LOG2": "10", "VKMR_BATCH_BYTES": "8192"}),
    (8, {"VKMR_SLICE_LOG2": "17", "VKMR_BATCH_MB": "2"}),        # 8 slices of 2^17, one per device (config 4's shape in small)
    (5, {"VKMR_SLICE_LOG2": "19", "VKMR_BATCH_MB": "4"}),        # fewer slices (2) than devices
    (2, {"VKMR_SLICE_LOG2": "8", "VKMR_BATCH_BYTES": "30000", "VKMR_SLICE_BUDGET": "1"}),
])
def test_hip_all_shards_slices_over_devices(native, golden, ndev, env):
    """"hip:all": slices round-robin over every device, per-device streams, batch and slice pools, one root array
    per device, ONE gather of the arrays, combine in slice order on the first device.  The box has one GPU: the
    LD_PRELOAD test double shows it k times to the unmodified binaries (see virtual_devices_env)."""
    env = dict(env, **virtual_devices_env(ndev))
    for name in ("G2_rndm_1712489279_1024_127", "G3_rndm_42_1048576_127", "G6_rndm_7_1000_300", "L7_three"):
        s = golden["streams"][name]
        r, out, m = run_vkmr(native, "hip:all", golden_stream(native, s), env)
        assert r.returncode == 0 and out[0] == "Initializing for: hip:all", (name, r.stderr[-400:])
        assert m and (m["name"], int(m["items"]), m["root"]) == ("hip:all", s["items"], s["root"]), (name, env)


@pytest.mark.gpu
def test_device_listing_with_several_devices(native):
    r, out, m = run_vkmr(native, None, b"a\n", virtual_devices_env(2))
    assert r.returncode == 1
    for want in (b"* CPU", b"* hip:0", b"* hip:1", b"* hip:all"):
        assert want in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("env", [
    {"VKMR_PACK_THREADS": "8", "VKMR_BATCH_MB": "16", "VKMR_SLICE_LOG2": "18"},
    {"VKMR_PACK_THREADS": "3", "VKMR_BATCH_MB": "64"},
    {"VKMR_PACK_THREADS": "16", "VKMR_INPUT_SPAN_MB": "4", "VKMR_BATCH_MB": "8", "VKMR_SLICE_LOG2": "19"},
    {"VKMR_PACK_THREADS": "1"},
    {"VKMR_DEVICE_SPLIT": "1"},                                                                   # the text split into strings on the device
    {"VKMR_DEVICE_SPLIT": "1", "VKMR_PACK_THREADS": "5", "VKMR_INPUT_SPAN_MB": "2", "VKMR_BATCH_MB": "3", "VKMR_SLICE_LOG2": "15"},
    {"VKMR_DEVICE_SPLIT": "1", "VKMR_PACK_THREADS": "7", "VKMR_INPUT_SPAN_MB": "8", "VKMR_BATCH_MB": "16", "VKMR_SLICE_LOG2": "19", "VKMR_MAX_INFLIGHT": "1"},
])
def test_hip_backend_mapped_file_and_parallel_packer(native, golden, tmp_path, env):
    """stdin redirected from a regular file (mapped) with the fork-join packer: spans cut at line ends,
    parts measured and packed in parallel, batch/slice boundaries falling inside spans."""
    s = golden["streams"]["G3_rndm_42_1048576_127"]
    path = tmp_path / "g3.txt"
    path.write_bytes(golden_stream(native, s))
    e = dict(os.environ)
    e.update(env)
    with open(path, "rb") as f:
        r = subprocess.run([tool(native, "vkmr"), "hip:0"], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)
    res = [LINE.match(l) for l in r.stdout.decode().splitlines()]
    m = [x for x in res if x][-1].groupdict()
    assert (int(m["items"]), int(m["bytes"]), m["root"]) == (s["items"], s["bytes"], s["root"]), env
    assert r.stderr.count(b"Read an empty string?") == 1


def test_cpu_backend_mapped_file(native, golden, tmp_path):
    s = golden["streams"]["G4_rndm_42_4096_4096"]
    path = tmp_path / "g4.txt"
    path.write_bytes(golden_stream(native, s))
    with open(path, "rb") as f:
        r = subprocess.run([tool(native, "vkmr"), "CPU"], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert s["root"] in r.stdout.decode()


def test_very_long_lines_pipe_and_mapped_file(native, oracle, tmp_path):
    """Lines longer than the reader's 16 MiB block (pipe: the buffer grows) and than the 32 MiB span of a
    mapped file (the span is extended to the line's end)."""
    data = b"head\n" + b"q" * (40 << 20) + b"\nmid\n" + b"r" * (70 << 20) + b"\ntail"
    want, cnt, nb = oracle.root_of_stream(data)
    r, out, m = run_vkmr(native, "CPU", data)
    assert m and (m["root"], int(m["items"]), int(m["bytes"])) == (want, cnt, nb)
    path = tmp_path / "long.txt"
    path.write_bytes(data)
    with open(path, "rb") as f:
        r = subprocess.run([tool(native, "vkmr"), "CPU"], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert want in r.stdout.decode() and f"(of {cnt} item(s), {nb} byte(s))" in r.stdout.decode()


@pytest.mark.gpu
def test_config3_from_stdin_matches_the_reference_cpu_path(native, tmp_path):
    """BASELINE configs[2] end to end, as the north star words it: `rndm 42 67108864 127 | vkmr hip:0` prints the root,
    the item count and the byte count that the reference's own CPU-serial path printed for the same stream
    (tests/golden/big_roots.json, made by oracle/_ref/rndm | oracle/_ref/vkmr_cpu_ref: four minutes there, a second here)."""
    import json
    from conftest import ROOT
    big = json.load(open(os.path.join(ROOT, "tests", "golden", "big_roots.json")))
    want = big["sub_roots"]["42"]
    path = tmp_path / "g26.txt"
    with open(path, "wb") as f:
        subprocess.check_call([tool(native, "rndm"), "42", str(big["count"]), str(big["maxlen"])], stdout=f, stderr=subprocess.DEVNULL)
    for env in ({}, {"VKMR_SLICE_LOG2": "26"}):     # eight slices of 2^23 (the reference's slice), and one slice
        e = dict(os.environ, **env)
        with open(path, "rb") as f:
            r = subprocess.run([tool(native, "vkmr"), "hip:0"], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=900)
        m = [x for x in (LINE.match(l) for l in r.stdout.decode().splitlines()) if x][-1].groupdict()
        assert (int(m["items"]), int(m["bytes"]), m["root"]) == (want["items"], want["bytes"], want["root"]), env
    os.unlink(path)


def _fold_proof(lines):
    d = lambda b: hashlib.sha256(hashlib.sha256(b).digest()).digest()
    cur = bytes.fromhex(lines[0].split()[-1])
    for l in lines[1:]:
        _, _, level, side, hexsib = l.split()
        cur = d(bytes.fromhex(hexsib) + cur) if side == "sibling-on-left" else d(cur + bytes.fromhex(hexsib))
    return cur.hex()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"VKMR_SLICE_LOG2": "16"}, {"VKMR_SLICE_LOG2": "13", "VKMR_SLICE_BUDGET": "2"}, {"VKMR_SLICE_LOG2": "17", "_ndev": "4"}])
def test_hip_merkle_proof_folds_to_the_root(native, golden, env):
    """VKMR_PROOF_INDEX on the GPU (the reference's to-do, README.md:118-120): the siblings inside a leaf's slice are written
    by that slice's reduction as it runs (vkmr_hip_reduce_proofs_async), those above by the combine of the slice roots;
    folding them with hashlib gives the root of the same run, which is the golden one.  One leaf per run, then all five --
    and one that is not in the stream -- in ONE run."""
    s = golden["streams"]["G3_rndm_42_1048576_127"]
    stream = golden_stream(native, s)
    leaves = [l for l in stream.split(b"\n") if l]
    env = dict(env)
    ndev = int(env.pop("_ndev", "1"))
    if ndev > 1:
        env.update(virtual_devices_env(ndev))
    for index in (0, 65535, 65536, 700001, 1048575):
        r, out, m = run_vkmr(native, "hip:all" if ndev > 1 else "hip:0", stream, dict(env, VKMR_PROOF_INDEX=str(index)))
        assert m and m["root"] == s["root"], (index, env, r.stderr[-300:])
        proof = [l for l in out if l.startswith("proof: ")]
        assert proof and proof[0].split()[2] == str(index)
        assert proof[0].split()[-1] == hashlib.sha256(hashlib.sha256(leaves[index]).digest()).digest().hex()
        assert _fold_proof(proof) == s["root"], (index, env)
    wanted = [0, 65535, 65536, 700001, 1048575, 1048576, 65536]
    r, out, m = run_vkmr(native, "hip:all" if ndev > 1 else "hip:0", stream, dict(env, VKMR_PROOF_INDEX=",".join(str(i) for i in wanted)))
    assert m and m["root"] == s["root"], (env, r.stderr[-300:])
    blocks = []
    for l in (l for l in out if l.startswith("proof: ")):
        if l.startswith("proof: leaf "):
            blocks.append([l])
        else:
            blocks[-1].append(l)
    assert [b[0].split()[2] for b in blocks] == [str(i) for i in wanted]
    for index, b in zip(wanted, blocks):
        if index >= len(leaves):
            assert "is not in the stream" in b[0] and len(b) == 1
            continue
        assert b[0].split()[-1] == hashlib.sha256(hashlib.sha256(leaves[index]).digest()).digest().hex()
        assert _fold_proof(b) == s["root"], (index, env)


@pytest.mark.gpu
def test_packed_pipeline_entry_point_on_the_gpu(native, golden):
    """vkmr_host_pipeline_packed (libvkmr_pipeline.so, what bench.py's pipeline_pcie_inclusive times): pre-packed strings
    through the C++ stream processor -- copy stream, map stream, reductions, combine -- against the golden root of
    rndm 42 2^20 127 (printed by the reference's CPU path), for a few batch and slice shapes."""
    import ctypes as C
    import vk_merkle_roots_amd as vk
    from vk_merkle_roots_amd.build import PIPELINE_LIB
    L = C.CDLL(PIPELINE_LIB)
    L.vkmr_host_pipeline_packed.restype = C.c_int
    L.vkmr_host_pipeline_packed.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_char_p, C.POINTER(C.c_double)]
    want = golden["streams"]["G3_rndm_42_1048576_127"]["root"]
    b = vk.rndm_packed(42, 1 << 20, 127)
    for per_batch, slice_log2 in ((1 << 20, 0), (1 << 17, 18), (100000, 16), (1 << 18, 20)):
        hexbuf, secs = C.create_string_buffer(65), C.c_double()
        rc = L.vkmr_host_pipeline_packed(0, b.data.ctypes.data, b.words, b.meta.ctypes.data, b.count, per_batch, slice_log2, hexbuf, C.byref(secs))
        assert rc == 0 and hexbuf.value.decode() == want, (per_batch, slice_log2, rc, hexbuf.value)
        assert 0 < secs.value < 5


@pytest.mark.gpu
def test_text_pipeline_entry_point_on_the_gpu(native, golden):
    """vkmr_host_pipeline_text (libvkmr_pipeline.so): the reference's run() as one call on text in memory -- the parallel
    packer, pinned batches, copy and map streams, reductions, combine, as `vkmr hip:0 < file` runs them -- against the
    golden root of rndm 42 2^20 127 (printed by the reference's CPU path), whole and in spans of 1 and 5 MiB."""
    import ctypes as C
    from vk_merkle_roots_amd.build import PIPELINE_LIB
    L = C.CDLL(PIPELINE_LIB)
    L.vkmr_host_pipeline_text.restype = C.c_int
    L.vkmr_host_pipeline_text.argtypes = [C.c_int, C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    s = golden["streams"]["G3_rndm_42_1048576_127"]
    text = golden_stream(native, s)
    for span in (0, 1 << 20, 5 << 20):
        hexbuf, items, nbytes, secs = C.create_string_buffer(65), C.c_uint64(), C.c_uint64(), C.c_double()
        rc = L.vkmr_host_pipeline_text(0, text, len(text), span, hexbuf, C.byref(items), C.byref(nbytes), C.byref(secs))
        assert (rc, hexbuf.value.decode(), items.value, nbytes.value) == (0, s["root"], s["items"], s["bytes"]), span
        assert 0 < secs.value < 5
