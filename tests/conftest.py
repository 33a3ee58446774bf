import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "vectors.json")
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_pattern(n, salt):
    """Same deterministic bytes as tests/golden/make_golden.py::pattern."""
    out = bytearray()
    x = (salt * 2654435761 + 12345) & 0xFFFFFFFF
    while len(out) < n:
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        b = (x >> 24) & 0xFF
        if b == 0x0A:
            b = 0x8A
        out.append(b)
    return bytes(out)


@pytest.fixture(scope="session")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


class Oracle:
    """ctypes view of oracle/liboracle.so -- the CHECKER.  Tests only."""

    def __init__(self):
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "sha256d_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "all"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.oracle_sha256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.oracle_leaf.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.oracle_node.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_root_inplace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_reduce_height.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p]
        L.oracle_words_to_hex.argtypes = [C.c_void_p, C.c_char_p]
        L.oracle_leaves_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.oracle_root_mt.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.oracle_root_of_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        self.L = L

    def sha256(self, msg):
        out = C.create_string_buffer(32)
        self.L.oracle_sha256(msg, len(msg), out)
        return out.raw

    def leaf(self, msg):
        out = np.zeros(8, dtype=np.uint32)
        self.L.oracle_leaf(msg, len(msg), out.ctypes.data)
        return out

    def node(self, l, r):
        l = np.ascontiguousarray(l, dtype=np.uint32)
        r = np.ascontiguousarray(r, dtype=np.uint32)
        out = np.zeros(8, dtype=np.uint32)
        self.L.oracle_node(l.ctypes.data, r.ctypes.data, out.ctypes.data)
        return out

    def root(self, leaves, threads=1):
        work = np.array(leaves, dtype=np.uint32, copy=True).reshape(-1, 8)
        out = np.zeros(8, dtype=np.uint32)
        if threads > 1:
            rc = self.L.oracle_root_mt(work.ctypes.data, work.shape[0], out.ctypes.data, threads)
        else:
            rc = self.L.oracle_root_inplace(work.ctypes.data, work.shape[0], out.ctypes.data)
        assert rc == 0
        return out

    def reduce_height(self, leaves, height):
        work = np.array(leaves, dtype=np.uint32, copy=True).reshape(-1, 8)
        out = np.zeros(8, dtype=np.uint32)
        assert self.L.oracle_reduce_height(work.ctypes.data, work.shape[0], height, out.ctypes.data) == 0
        return out

    def hex(self, words):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        buf = C.create_string_buffer(65)
        self.L.oracle_words_to_hex(words.ctypes.data, buf)
        return buf.value.decode()

    def leaves_packed(self, data, meta, threads=1):
        data = np.ascontiguousarray(data, dtype=np.uint32)
        meta = np.ascontiguousarray(meta, dtype=np.uint32)
        n = meta.shape[0]
        out = np.zeros((n, 8), dtype=np.uint32)
        if n:
            self.L.oracle_leaves_packed(data.ctypes.data, meta.ctypes.data, n, out.ctypes.data, threads)
        return out

    def root_of_stream(self, stream):
        hexbuf = C.create_string_buffer(65)
        cnt, nb = C.c_uint64(0), C.c_uint64(0)
        rc = self.L.oracle_root_of_stream(stream, len(stream), hexbuf, C.byref(cnt), C.byref(nb))
        return (hexbuf.value.decode() if rc == 0 else ""), cnt.value, nb.value


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def ref_lib():
    """The reference's own CPU path compiled into oracle/_ref (built where /root/reference
    exists; travels prebuilt to the GPU box).  Skips when absent."""
    so = os.path.join(REF_DIR, "libvkmr_ref.so")
    if not os.path.exists(so):
        if os.path.isdir("/root/reference/src/vkmr"):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
        else:
            pytest.skip("oracle/_ref not built and /root/reference absent")
    L = C.CDLL(so)
    L.ref_sha256d.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    L.ref_sha256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    L.ref_root.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_char_p]
    return L


@pytest.fixture(scope="session")
def native():
    """Built native parts (HIP extension cross-compiles without a GPU)."""
    from vk_merkle_roots_amd import build
    build.build_all()
    return build


def build_virt_devices():
    """tests/_build/libvirt_devices.so: the LD_PRELOAD test double that shows one GPU as several
    (tests/c/virt_devices.cpp).  Test infrastructure; never part of the product."""
    src = os.path.join(ROOT, "tests", "c", "virt_devices.cpp")
    out = os.path.join(ROOT, "tests", "_build", "libvirt_devices.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-shared", "-fPIC", src, "-o", out,
                               "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath,/opt/rocm/lib"])
    return out


@pytest.fixture(scope="session")
def gpu(native):
    import vk_merkle_roots_amd as vk
    return vk.HipDevice(0)
