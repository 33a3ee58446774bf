"""H2D rate from pinned memory: one copy stream against two in parallel (does a second SDMA engine add anything on this link?).
    python3 tools/h2d_probe.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vk_merkle_roots_amd as vk  # noqa: E402

dev = vk.HipDevice(0)
L = dev.lib
N = 256 << 20
hosts, devs = [], []
for _ in range(2):
    h = C.c_void_p()
    vk.check(L.vkmr_hip_host_alloc(N, C.byref(h)), "host_alloc")
    C.memset(h, 1, N)
    hosts.append(h)
    devs.append(dev.alloc(N))
s1, s2 = dev.new_stream(), dev.new_stream()
dev.warm_up(kernels=False, copy_bytes=1 << 20, stream=s1)
dev.warm_up(kernels=False, copy_bytes=1 << 20, stream=s2)


def copy(stream, k, off, n):
    vk.check(L.vkmr_hip_memcpy_h2d_async(dev.index, stream, devs[k].at(off), hosts[k].value + off, n), "h2d")


for chunk in (40 << 20, 8 << 20, 1 << 20):
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(2):
            for off in range(0, N, chunk):
                copy(s1, k, off, min(chunk, N - off))
        dev.sync(s1)
        one = time.perf_counter() - t0
        t0 = time.perf_counter()
        for off in range(0, N, chunk):
            copy(s1, 0, off, min(chunk, N - off))
            copy(s2, 1, off, min(chunk, N - off))
        dev.sync(s1)
        dev.sync(s2)
        two = time.perf_counter() - t0
        print(f"copies of {chunk >> 20} MiB, 512 MiB in all: one stream {2 * N / one / 1e9:.1f} GB/s, two streams in parallel {2 * N / two / 1e9:.1f} GB/s", flush=True)
