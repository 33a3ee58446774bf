"""The reduction schedule never writes more scratch than the size functions promise
(csrc/reduce_plan.hpp): a CPU-side replay over many counts, short last slices and the
sibling sub-trees of proofs (tests/c/reduce_plan_test.cpp), plus the same property seen
through the C ABI's size functions.  No GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_schedule_replay_stays_within_scratch_budget(tmp_path):
    exe = str(tmp_path / "reduce_plan_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "vk_merkle_roots_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "reduce_plan_test.cpp"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert b"ok:" in r.stdout


def test_scratch_size_functions_are_monotone_upper_bounds(native):
    """vkmr_hip_reduce_scratch_bytes(count) must cover every shorter run: it may not shrink when count grows,
    and the slices form must cover the one-slice form of any shorter last slice."""
    import vk_merkle_roots_amd as vk
    L = vk.lib()
    prev = 0
    for lg in range(0, 34):
        for n in ((1 << lg) - 1, 1 << lg, (1 << lg) + 1, 3 << lg):
            if n < 1:
                continue
            b = L.vkmr_hip_reduce_scratch_bytes(n)
            assert b % 32 == 0 and b >= 64
            if n >= prev:
                assert b >= L.vkmr_hip_reduce_scratch_bytes(max(1, prev)), (n, prev)
            prev = max(prev, n)
    # the ADVICE r1 case: capacity 2^20, one slice holding 2^20 - 256 nodes needs 786240 cells
    assert L.vkmr_hip_reduce_slices_scratch_bytes(1 << 20, 1) >= 786240 * 32
    assert L.vkmr_hip_reduce_scratch_bytes(3 * (1 << 20) - 256) >= 786240 * 32
    # refused: heights that cannot be the height of any tree
    assert L.vkmr_hip_reduce_async(0, None, 1, 5, 64, None, 1) == vk._abi.ERR_INVALID
