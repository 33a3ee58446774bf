"""The C-ABI library loads and exports every symbol include/vkmr_hip.h declares.
No compute calls here: this runs without a GPU."""
import ctypes as C
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vkmr_hip.h")).read()
    return sorted(set(re.findall(r"VKMR_API\s+[\w\s\*]+?\b(vkmr_hip_\w+)\s*\(", text)))


def test_header_declares_the_hot_path():
    names = declared_symbols()
    for must in ("vkmr_hip_map_async", "vkmr_hip_reduce_async", "vkmr_hip_combine_async", "vkmr_hip_gather_roots_async", "vkmr_hip_event_query",
                 "vkmr_hip_device_count", "vkmr_hip_host_alloc", "vkmr_hip_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(native):
    lib = C.CDLL(native.HIP_LIB)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_python_stub_covers_the_header(native):
    from vk_merkle_roots_amd import _abi
    assert sorted(_abi.SIGNATURES) == declared_symbols()
    _abi.lib()   # binds every signature; raises if one is missing


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "vkmr_hip.h"\nint main(void){ vkmr_digest d; vkmr_metadata m; (void)d; (void)m; '
                   'return sizeof(vkmr_digest) == 32 && sizeof(vkmr_metadata) == 8 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(exe) + ".o"])


def test_no_gpu_calls_fail_cleanly(native):
    """Without a device the ABI reports it instead of crashing (skipped on a GPU box)."""
    from vk_merkle_roots_amd import _abi
    lib = _abi.lib()
    n = C.c_int(-1)
    assert lib.vkmr_hip_device_count(C.byref(n)) == 0
    if n.value > 0:
        return
    p = C.c_void_p()
    assert lib.vkmr_hip_device_alloc(0, 1024, C.byref(p)) < 0
    assert lib.vkmr_hip_last_error()


def test_invalid_arguments_are_rejected(native):
    from vk_merkle_roots_amd import _abi
    lib = _abi.lib()
    assert lib.vkmr_hip_device_count(None) == _abi.ERR_INVALID
    assert lib.vkmr_hip_reduce_async(0, None, None, 4, 2, None, None) == _abi.ERR_INVALID
    # height must reduce count to one node
    dummy = C.c_void_p(0x1000)
    assert lib.vkmr_hip_reduce_async(0, None, dummy, 5, 2, dummy, dummy) == _abi.ERR_INVALID
    assert lib.vkmr_hip_reduce_scratch_bytes(1 << 23) < (1 << 23) * 32 // 8     # scratch is a small fraction of the slice
    assert lib.vkmr_hip_reduce_scratch_bytes(1 << 26) < (1 << 26) * 32 // 8      # a pass collapses 4 levels: 1/16 + 1/32 of the slice


def test_product_does_not_reference_the_oracle():
    """The oracle is the checker only: nothing under the package or include/ may mention it."""
    bad = []
    for base in ("vk_merkle_roots_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", ".c")):
                    t = open(os.path.join(d, f), errors="replace").read()
                    if re.search(r"liboracle|oracle_|oracle/|_ref/|hashlib", t):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_product_does_not_reference_the_test_doubles():
    """tests/c/fake_vkmr_hip.cpp (host memory + CPU hashing behind the C ABI) and tests/c/virt_devices.cpp (one GPU shown
    as several) exist for the test suites only: nothing the product ships may name or load them."""
    bad = []
    for base in ("vk_merkle_roots_amd", "include", "bench.py", "__graft_entry__.py"):
        path = os.path.join(ROOT, base)
        files = [path] if os.path.isfile(path) else [os.path.join(d, f) for d, _, fs in os.walk(path) for f in fs
                                                     if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", ".c"))]
        for f in files:
            t = open(f, errors="replace").read()
            if re.search(r"fake_vkmr|virt_devices|tests/_build|VKMR_FAKE_|VKMR_TEST_VIRTUAL", t):
                bad.append(f)
    assert not bad, bad


def test_docs_name_only_entry_points_that_exist():
    """INTEGRATION.md, DESIGN.md and README.md may only name `vkmr_hip_*` entry points the header declares (or host-library
    helpers, `vkmr_host_*`): stale names in the binding guide would send a maintainer looking for functions that are gone."""
    declared = set(declared_symbols())
    # the experiments build's entry points (include/vkmr_hip_experiments.h) may be named too -- as what they are
    declared |= set(re.findall(r"VKMR_API\s+[\w\s\*]+?\b(vkmr_hip_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "vkmr_hip_experiments.h")).read()))
    for doc in ("INTEGRATION.md", "DESIGN.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for name in set(re.findall(r"\b(vkmr_hip_[a-z0-9_]+)\b", text)):
            if name in ("vkmr_hip_h", "vkmr_hip_experiments", "vkmr_hip_comm_", "vkmr_hip_combine"):   # file name fragments / prefix mention / round-1 name discussed in DESIGN
                continue
            assert name in declared or name + "_async" in declared or name.rstrip("_") in {d[: len(name.rstrip("_"))] for d in declared}, (doc, name)


def test_the_product_library_carries_no_experiment_knobs(native):
    """the A/B knobs (VKMR_MAP_VARIANT / _FIT / _TILE / _DYNLDS) and the non-shipped map_kernel instantiations
    live in the experiments build only; the product picks its fetch mode from the batch alone and ships five
    instantiations (LDS-staged tiles; per-lane loads, one block per trip and two, each with 512 and with 256 lanes)."""
    import subprocess
    from vk_merkle_roots_amd import build
    blob = open(build.HIP_LIB, "rb").read()
    for knob in (b"VKMR_MAP_VARIANT", b"VKMR_MAP_FIT", b"VKMR_MAP_TILE", b"VKMR_MAP_DYNLDS"):
        assert knob not in blob, knob
    syms = subprocess.run(["nm", "-C", build.HIP_LIB], stdout=subprocess.PIPE).stdout.decode()
    kernels = sorted({l.split(" V ")[1] for l in syms.splitlines() if " V void map_kernel<" in l})
    assert len(kernels) == 5, kernels
    assert any("map_kernel<512, 1024, 17664, 0, false, 0>" in k for k in kernels)
    if os.path.exists(build.EXP_LIB):
        assert b"VKMR_MAP_VARIANT" in open(build.EXP_LIB, "rb").read()
