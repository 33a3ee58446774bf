"""bench.py's orchestration on a CPU-only machine: the JSON contract, the slice/height/combine logic, the self-spawned
N-rank launch and its labelled gloo rehearsal -- driven through the test double of the C ABI (tests/c/fake_vkmr_hip.cpp,
selected with VKMR_HIP_LIB, the same switch tools/ab.sh uses for A/B builds).  Hashing in the double is done by the
product's own CPU functions, so the roots are real and are compared with the oracle; the timings mean nothing."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "vk_merkle_roots_amd", "csrc", "host")
CSRC = os.path.join(ROOT, "vk_merkle_roots_amd", "csrc")
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "root"]


@pytest.fixture(scope="session")
def fake_lib(native):
    out = os.path.join(ROOT, "tests", "_build", "fake_plain")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libvkmr_hip.so")
    srcs = [os.path.join(ROOT, "tests", "c", "fake_vkmr_hip.cpp"), os.path.join(HOST, "cpu_sha256d.cpp")]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs + [os.path.join(CSRC, "reduce_plan.hpp")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", HOST, "-I", CSRC] + srcs + ["-o", lib])
    return lib


def run_bench(fake_lib, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["VKMR_HIP_LIB"] = fake_lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-pipeline", "--no-long-strings"] + list(args),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    return r


def oracle_root(oracle, seed, n):
    import vk_merkle_roots_amd as vk
    b = vk.rndm_packed(seed, n, 127)
    return oracle.root(oracle.leaves_packed(b.data, b.meta, threads=4), threads=4)


@pytest.mark.parametrize("extra", [[], ["--slice-log2", "10"], ["--batch-log2", "11"], ["--slice-log2", "9", "--batch-log2", "12", "--levels-variant"]])
def test_bench_line_contract_and_root(fake_lib, oracle, extra):
    r = run_bench(fake_lib, "--leaves-log2", "13", "--steps", "2", "--warmup", "1", *extra)
    assert r.returncode == 0, r.stderr[-1500:].decode()
    lines = r.stdout.decode().strip().splitlines()
    assert len(lines) == 1, "stdout must carry the one JSON line and nothing else"
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and "workload" in d["config"]
    assert "roofline_reduce" in d and d["config"]["leaves_per_gpu"] == 1 << 13
    assert d["root"] == oracle.hex(oracle_root(oracle, 42, 1 << 13))
    assert d["sub_roots"] == [d["root"]] and d["root_matches_golden"] is None      # the golden file covers 2^26 only


@pytest.mark.parametrize("n", [2, 3])
def test_bench_spawns_its_ranks_and_combines_in_rank_order(fake_lib, oracle, n):
    """`python bench.py --gpus N --rehearse-gloo`: the script starts the N ranks itself (no launcher), rank r hashes
    rndm 42+r, the sub-roots are gathered in rank order and combined with the duplicate-last rule."""
    r = run_bench(fake_lib, "--gpus", str(n), "--rehearse-gloo", "--leaves-log2", "12", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-1500:].decode()
    lines = r.stdout.decode().strip().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["ranks"] == n and "REHEARSAL" in d["config"]["collective"]
    subs = np.stack([oracle_root_words for oracle_root_words in (oracle.reduce_height(_leaves(oracle, 42 + k, 1 << 12), 12) for k in range(n))])
    assert d["sub_roots"] == [oracle.hex(s) for s in subs]
    assert d["root"] == oracle.hex(oracle.root(subs))
    assert d["value"] > 0 and d["config"]["leaves_total"] == n << 12


def _leaves(oracle, seed, n):
    import vk_merkle_roots_amd as vk
    b = vk.rndm_packed(seed, n, 127)
    return oracle.leaves_packed(b.data, b.meta, threads=4)


def test_bench_secondary_blocks_agree_with_the_timed_root(fake_lib, oracle):
    """pipeline_pcie_inclusive (pinned host batches, copy stream) and two_stream_overlap (map and reduce streams, two
    digest buffers) are separate drivings of the same ABI: each must reproduce the timed path's root."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["VKMR_HIP_LIB"] = fake_lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--leaves-log2", "12", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-long-strings"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:].decode()
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["root"] == oracle.hex(oracle_root(oracle, 42, 1 << 12))
    assert d["pipeline_pcie_inclusive"]["root_matches"] is True
    assert d["two_stream_overlap"]["roots_match"] is True


def test_an_n_rank_line_is_as_complete_as_the_one_rank_line(fake_lib, oracle):
    """with N > 1 rank 0 still reports `roofline`, `valu_roofline` and `cpu_baseline` (the reference's CPU
    path on rank 0's stream, labelled as 1/N of the workload) -- after every rank has left the data path.  Eight ranks, the
    script's own launcher, the labelled gloo rehearsal through the fake ABI."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["VKMR_HIP_LIB"] = fake_lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rehearse-gloo", "--leaves-log2", "10", "--steps", "1", "--warmup", "1",
                        "--no-pipeline", "--no-long-strings"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:].decode()
    lines = r.stdout.decode().strip().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in REQUIRED + ["roofline_reduce", "valu_roofline", "cpu_baseline"]:
        assert k in d, k
    assert d["n_gpus"] == 8 and d["config"]["ranks"] == 8 and d["config"]["leaves_total"] == 8 << 10 and len(d["sub_roots"]) == 8
    cb = d["cpu_baseline"]
    assert cb["cores"] == 1 and "1/8 of the workload" in cb["sample"] and cb["leaves"] == 1 << 10
    assert cb["root_matches_gpu"] is True                      # rank 0's sub-tree root IS the root of rank 0's stream
    subs = np.stack([oracle.reduce_height(_leaves(oracle, 42 + k, 1 << 10), 10) for k in range(8)])
    assert d["root"] == oracle.hex(oracle.root(subs))


def test_what_an_eight_gpu_line_says_about_its_gather_and_its_golden_roots():
    """The parts of an 8-GPU line that cannot run here (RCCL, 2^29 leaves), as the functions that write them: the gather is
    named as ONE RCCL all-gather through the C ABI, the workload as BASELINE configs[3], and the root is checked against the
    combination of the eight reference-made sub-roots in rank order (tests/golden/big_roots.json) -- a wrong order or a wrong
    sub-root is reported as such."""
    sys.path.insert(0, ROOT)
    import bench
    assert "ncclAllGather" in bench.gather_label(True, True) and "C ABI" in bench.gather_label(True, True)
    assert "gloo" in bench.gather_label(True, False) and bench.gather_label(False, False) is None
    assert "configs[3]" in bench.workload_label(42, 26, 127, 8) and "configs[2]" in bench.workload_label(42, 26, 127, 1)
    golden = bench.golden_big_roots(26, 127)
    assert golden and set(golden["combined"]) >= {str(k) for k in range(1, 9)}
    subs = [golden["sub_roots"][str(42 + r)]["root"] for r in range(8)]
    assert bench.check_against_golden(golden, 42, 8, golden["combined"]["8"], subs) == (True, True)
    assert bench.check_against_golden(golden, 42, 8, golden["combined"]["8"], subs[::-1]) == (True, False)
    assert bench.check_against_golden(golden, 42, 8, golden["combined"]["4"], subs) == (False, True)
    assert bench.check_against_golden(golden, 42, 4, golden["combined"]["4"], subs[:4]) == (True, True)
    assert bench.check_against_golden(None, 42, 8, "", []) == (None, None)


def test_the_golden_file_pins_config5():
    """bench.py's config5_full leg checks its root against what the reference's own CPU path printed for `rndm 42 2^24 4096`
    (BASELINE configs[4] at full size: 34 GB of text, 38 minutes of the reference; tests/golden/make_big_roots.py --config5)."""
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "big_roots.json")))["config5"]
    assert rec["generator"] == "rndm 42 16777216 4096" and rec["items"] == 1 << 24 and len(rec["root"]) == 64
    assert rec["bytes"] > 34e9
