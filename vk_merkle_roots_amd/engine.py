"""Thin Python driver over the C ABI, used by tests/ and bench.py.

Mirrors, call for call, what the C++ stream processor (csrc/host/hip_sha256d.cpp) does
with the same ABI: upload a packed batch, map it into a slice of digests, reduce each
slice to `height` levels, combine the slice roots.  All hashing happens on the GPU.
"""
import ctypes as C
import math

import numpy as np

from . import _abi
from ._abi import check


def tree_height(count):
    """Levels of the duplicate-last tree over `count` leaves; a lone leaf is still
    hashed with itself once (CpuSha256D::Root's do-while, reference
    src/vkmr/SHA-256plus.cpp:515-547)."""
    if count <= 0:
        raise ValueError("count must be positive")
    return max(1, int(count - 1).bit_length())


class PackedBatch:
    """Strings packed as the reference's Batch does (src/vkmr/Batches.cpp:64-121):
    `data` uint32 words, `meta` uint32 [count, 2] = {start word, size bytes}."""

    def __init__(self, data, meta, words, nbytes):
        self.data = data
        self.meta = meta
        self.count = int(meta.shape[0])
        self.words = int(words)
        self.nbytes = int(nbytes)

    def slice(self, lo, hi):
        """Strings [lo, hi) as their own batch (metadata rebased to word 0)."""
        meta = self.meta[lo:hi].copy()
        if hi <= lo:
            return PackedBatch(self.data[:0], meta, 0, 0)
        w0 = int(meta[0, 0])
        w1 = int(meta[-1, 0]) + (int(meta[-1, 1]) + 3) // 4
        meta[:, 0] -= np.uint32(w0)
        return PackedBatch(self.data[w0:w1], meta, w1 - w0, int(meta[:, 1].astype(np.uint64).sum()))


def pack_lines(stream, data_capacity_words=None):
    """Split `stream` (bytes) with the reference's line rules and pack the lines."""
    h = _abi.host_lib()
    buf = np.frombuffer(stream, dtype=np.uint8)
    n = len(stream)
    max_count = n // 2 + 2
    cap = data_capacity_words or (n // 4 + max_count + 4)
    data = np.zeros(cap, dtype=np.uint32)
    meta = np.zeros((max_count, 2), dtype=np.uint32)
    words = C.c_uint64(0)
    nbytes = C.c_uint64(0)
    cnt = h.vkmr_host_pack_lines(buf.ctypes.data if n else None, n, data.ctypes.data, cap, meta.ctypes.data, max_count,
                                 C.byref(words), C.byref(nbytes))
    if cnt < 0:
        raise RuntimeError("pack_lines: buffers too small")
    return PackedBatch(data[: words.value], meta[:cnt], words.value, nbytes.value)


def rndm_packed(seed, count, maxlen):
    """The strings of `rndm seed count maxlen`, generated straight into a packed batch."""
    h = _abi.host_lib()
    cap = int(count) * ((maxlen - 2) // 4 + 1) + 4
    data = np.empty(cap, dtype=np.uint32)           # untouched pages cost nothing
    meta = np.empty((count, 2), dtype=np.uint32)
    words = C.c_uint64(0)
    cnt = h.vkmr_host_rndm_pack(seed, count, maxlen, data.ctypes.data, cap, meta.ctypes.data, C.byref(words))
    if cnt != count:
        raise RuntimeError("rndm_packed: buffer too small")
    return PackedBatch(data[: words.value], meta, words.value, int(meta[:, 1].astype(np.uint64).sum()))


class RndmStream:
    """The stream of `rndm seed * maxlen`, handed out as consecutive packed batches."""

    def __init__(self, seed, maxlen):
        self.h = _abi.host_lib()
        self.maxlen = maxlen
        self.handle = self.h.vkmr_host_rndm_open(seed)

    def next(self, count):
        cap = int(count) * ((self.maxlen - 2) // 4 + 1) + 4
        data = np.empty(cap, dtype=np.uint32)
        meta = np.empty((count, 2), dtype=np.uint32)
        words = C.c_uint64(0)
        if self.h.vkmr_host_rndm_next(self.handle, count, self.maxlen, data.ctypes.data, cap, meta.ctypes.data, C.byref(words)) != count:
            raise RuntimeError("RndmStream.next: batch exceeds 2^32 words")
        return PackedBatch(data[: words.value], meta, words.value, int(meta[:, 1].astype(np.uint64).sum()))

    def close(self):
        if self.handle:
            self.h.vkmr_host_rndm_close(self.handle)
            self.handle = None


class DeviceBuffer:
    def __init__(self, dev, nbytes):
        self.dev = dev
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(dev.lib.vkmr_hip_device_alloc(dev.index, max(self.nbytes, 32), C.byref(p)), "vkmr_hip_device_alloc")
        self.ptr = p.value

    def at(self, byte_offset):
        return self.ptr + int(byte_offset)

    def free(self):
        if self.ptr:
            self.dev.lib.vkmr_hip_device_free(self.dev.index, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HipDevice:
    """One GPU + one stream.  Every method is one or two ABI calls."""

    def __init__(self, index=0):
        self.lib = _abi.lib()
        self.index = index
        n = C.c_int(0)
        check(self.lib.vkmr_hip_device_count(C.byref(n)), "vkmr_hip_device_count")
        if index >= n.value:
            raise RuntimeError(f"HIP device {index} not present ({n.value} device(s)): {_abi.what_error()}")
        s = C.c_void_p()
        check(self.lib.vkmr_hip_stream_create(index, C.byref(s)), "vkmr_hip_stream_create")
        self.stream = s.value

    # -- plumbing -----------------------------------------------------------------
    def name(self):
        buf = C.create_string_buffer(256)
        check(self.lib.vkmr_hip_device_name(self.index, buf, 256), "vkmr_hip_device_name")
        return buf.value.decode()

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def upload(self, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        buf = DeviceBuffer(self, arr.nbytes)
        if arr.nbytes:
            check(self.lib.vkmr_hip_memcpy_h2d_async(self.index, stream or self.stream, buf.ptr, arr.ctypes.data, arr.nbytes),
                  "vkmr_hip_memcpy_h2d_async")
            self.sync(stream)
        return buf

    def download(self, buf, nbytes, dtype=np.uint32, offset=0, stream=None):
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        if nbytes:
            check(self.lib.vkmr_hip_memcpy_d2h_async(self.index, stream or self.stream, out.ctypes.data, buf.at(offset), nbytes),
                  "vkmr_hip_memcpy_d2h_async")
            self.sync(stream)
        return out

    def sync(self, stream=None):
        check(self.lib.vkmr_hip_stream_sync(self.index, stream or self.stream), "vkmr_hip_stream_sync")

    def new_stream(self):
        s = C.c_void_p()
        check(self.lib.vkmr_hip_stream_create(self.index, C.byref(s)), "vkmr_hip_stream_create")
        return s.value

    def warm_up(self, kernels=True, copy_bytes=1 << 20, stream=None):
        """vkmr_hip_warm_up: the kernels loaded onto the device, the copy engine up -- what a caller does at start-up
        instead of inside its first copy and first launch (the reference builds its pipelines then: Devices.cpp:225-280)."""
        what = (1 if kernels else 0) | (2 if copy_bytes else 0)
        check(self.lib.vkmr_hip_warm_up(self.index, stream or self.stream, what, copy_bytes or 0), "vkmr_hip_warm_up")

    def new_event(self):
        e = C.c_void_p()
        check(self.lib.vkmr_hip_event_create(self.index, C.byref(e)), "vkmr_hip_event_create")
        return e.value

    def record(self, event, stream=None):
        check(self.lib.vkmr_hip_event_record(self.index, event, stream or self.stream), "vkmr_hip_event_record")

    def wait(self, event):
        check(self.lib.vkmr_hip_event_wait(self.index, event), "vkmr_hip_event_wait")

    def elapsed_ms(self, e0, e1):
        ms = C.c_float(0)
        check(self.lib.vkmr_hip_event_elapsed_ms(self.index, e0, e1, C.byref(ms)), "vkmr_hip_event_elapsed_ms")
        return ms.value

    # -- the hot path -------------------------------------------------------------
    def map_async(self, data_buf, data_words, meta_buf, count, out_buf, out_offset_digests=0, meta_offset=0, stream=None):
        check(self.lib.vkmr_hip_map_async(self.index, stream or self.stream, data_buf.ptr, data_words,
                                          meta_buf.at(8 * meta_offset), count, out_buf.at(32 * out_offset_digests)),
              "vkmr_hip_map_async")

    def reduce_async(self, digests_buf, count, height, scratch_buf, root_buf, root_index=0, levels_variant=False, stream=None):
        fn = self.lib.vkmr_hip_reduce_levels_async if levels_variant else self.lib.vkmr_hip_reduce_async
        check(fn(self.index, stream or self.stream, digests_buf.ptr, count, height, scratch_buf.ptr if scratch_buf else None,
                 root_buf.at(32 * root_index)),
              "vkmr_hip_reduce_async")

    def reduce_slices_async(self, digests_buf, nslices, capacity, count_last, height, scratch_buf, roots_buf, stream=None):
        check(self.lib.vkmr_hip_reduce_slices_async(self.index, stream or self.stream, digests_buf.ptr, nslices, capacity, count_last,
                                                    height, scratch_buf.ptr if scratch_buf else None, roots_buf.ptr),
              "vkmr_hip_reduce_slices_async")

    def proof(self, digests_buf, count, height, index):
        """(siblings [height, 8], root [8]) of leaf `index` in the tree reduce_async(count, height) computes."""
        d_sib = self.alloc(32 * max(height, 1))
        d_root = self.alloc(32)
        d_scratch = self.reduce_scratch(count)
        check(self.lib.vkmr_hip_proof_async(self.index, self.stream, digests_buf.ptr, count, height, index, d_scratch.ptr, d_sib.ptr,
                                            d_root.ptr), "vkmr_hip_proof_async")
        sib = self.download(d_sib, 32 * height).reshape(-1, 8) if height else np.zeros((0, 8), np.uint32)
        root = self.download(d_root, 32)
        for b in (d_sib, d_root, d_scratch):
            b.free()
        return sib, root

    def reduce_with_proofs(self, digests_buf, count, height, indices):
        """(siblings [k, height, 8], root [8]): the reduction that writes the proofs of leaves `indices` while it runs
        (vkmr_hip_reduce_proofs_async; k <= 16)."""
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        k = int(idx.shape[0])
        d_sib = self.alloc(32 * max(height, 1) * max(k, 1))
        d_root = self.alloc(32)
        d_scratch = self.reduce_scratch(count)
        check(self.lib.vkmr_hip_reduce_proofs_async(self.index, self.stream, digests_buf.ptr, count, height, d_scratch.ptr, d_root.ptr,
                                                    idx.ctypes.data if k else None, k, d_sib.ptr), "vkmr_hip_reduce_proofs_async")
        sib = self.download(d_sib, 32 * height * k).reshape(k, height, 8) if height and k else np.zeros((k, height, 8), np.uint32)
        root = self.download(d_root, 32)
        for b in (d_sib, d_root, d_scratch):
            b.free()
        return sib, root

    def reduce_scratch(self, count, levels_variant=False):
        fn = self.lib.vkmr_hip_reduce_levels_scratch_bytes if levels_variant else self.lib.vkmr_hip_reduce_scratch_bytes
        return self.alloc(fn(count))

    def combine_async(self, roots_buf, n, scratch_buf, root_buf, stream=None):
        check(self.lib.vkmr_hip_combine_async(self.index, stream or self.stream, roots_buf.ptr, n, scratch_buf.ptr if scratch_buf else None,
                                              root_buf.ptr), "vkmr_hip_combine_async")

    def combine(self, roots):
        """Root ([8] uint32) over slice roots given as a host array, in slice order."""
        roots = np.ascontiguousarray(roots, dtype=np.uint32).reshape(-1, 8)
        d_in, d_root = self.upload(roots), self.alloc(32)
        d_scratch = self.reduce_scratch(roots.shape[0])
        self.combine_async(d_in, roots.shape[0], d_scratch, d_root)
        out = self.download(d_root, 32)
        for b in (d_in, d_root, d_scratch):
            b.free()
        return out

    # -- conveniences used by tests ------------------------------------------------
    def leaf_digests(self, batch):
        """Digests of every string of `batch` as a [count, 8] uint32 array."""
        if batch.count == 0:
            return np.zeros((0, 8), dtype=np.uint32)
        d_data = self.upload(batch.data if batch.words else np.zeros(1, np.uint32))
        d_meta = self.upload(batch.meta)
        d_out = self.alloc(32 * batch.count)
        self.map_async(d_data, batch.words, d_meta, batch.count, d_out)
        out = self.download(d_out, 32 * batch.count).reshape(-1, 8)
        for b in (d_data, d_meta, d_out):
            b.free()
        return out

    def reduce_digests(self, digests, height=None, levels_variant=False):
        """Sub-tree root ([8] uint32) of a [count, 8] uint32 array of digests."""
        digests = np.ascontiguousarray(digests, dtype=np.uint32).reshape(-1, 8)
        count = digests.shape[0]
        if height is None:
            height = tree_height(count)
        d_in = self.upload(digests)
        d_scratch = self.reduce_scratch(count, levels_variant)
        d_root = self.alloc(32)
        self.reduce_async(d_in, count, height, d_scratch, d_root, levels_variant=levels_variant)
        root = self.download(d_root, 32)
        for b in (d_in, d_scratch, d_root):
            b.free()
        return root


def digest_hex(words):
    """Canonical hex of a word-valued digest (big-endian bytes of H[0..7])."""
    return np.ascontiguousarray(words, dtype=np.uint32).astype(">u4").tobytes().hex()


def merkle_root_packed_batched(dev, batch, slice_capacity, batch_strings=None):
    """Same tree as merkle_root_packed, all slices resident and reduced by ONE
    vkmr_hip_reduce_slices_async, slice roots combined on the device."""
    n = batch.count
    if n == 0:
        return ""
    nslices = (n + slice_capacity - 1) // slice_capacity
    batch_strings = batch_strings or n
    d_slices = dev.alloc(32 * nslices * slice_capacity)
    for b0 in range(0, n, batch_strings):
        b1 = min(n, b0 + batch_strings)
        sub = batch.slice(b0, b1)
        d_data = dev.upload(sub.data if sub.words else np.zeros(1, np.uint32))
        d_meta = dev.upload(sub.meta)
        dev.map_async(d_data, sub.words, d_meta, sub.count, d_slices, out_offset_digests=b0)
        dev.sync()
        d_data.free()
        d_meta.free()
    count_last = n - (nslices - 1) * slice_capacity
    height = int(math.log2(slice_capacity)) if nslices > 1 else tree_height(n)
    d_scratch = dev.alloc(dev.lib.vkmr_hip_reduce_slices_scratch_bytes(slice_capacity, nslices))
    d_roots = dev.alloc(32 * nslices)
    dev.reduce_slices_async(d_slices, nslices, slice_capacity, count_last, height, d_scratch, d_roots)
    if nslices > 1:
        d_top = dev.reduce_scratch(nslices)
        d_final = dev.alloc(32)
        dev.reduce_async(d_roots, nslices, tree_height(nslices), d_top, d_final)
        root = dev.download(d_final, 32)
    else:
        root = dev.download(d_roots, 32)
    return digest_hex(root)


def merkle_root_packed(dev, batch, slice_capacity=None, batch_strings=None, levels_variant=False):
    """Root (hex) of all strings of `batch`: map in sub-batches of `batch_strings`,
    slices of `slice_capacity` digests (a power of two), per-slice reduce, combine.
    Same decomposition as the reference's stream processor (src/vkmr/SHA-256vk.cpp:288-429);
    by SURVEY.md 8a Q6 the result equals the single global duplicate-last tree."""
    n = batch.count
    if n == 0:
        return ""
    if slice_capacity is None:
        slice_capacity = 1 << max(1, (n - 1).bit_length())
    if slice_capacity & (slice_capacity - 1):
        raise ValueError("slice capacity must be a power of two")
    batch_strings = batch_strings or n
    nslices = (n + slice_capacity - 1) // slice_capacity
    cap_height = int(math.log2(slice_capacity))
    d_roots = dev.alloc(32 * nslices)
    for s in range(nslices):
        lo, hi = s * slice_capacity, min(n, (s + 1) * slice_capacity)
        d_slice = dev.alloc(32 * (hi - lo))
        for b0 in range(lo, hi, batch_strings):
            b1 = min(hi, b0 + batch_strings)
            sub = batch.slice(b0, b1)
            d_data = dev.upload(sub.data if sub.words else np.zeros(1, np.uint32))
            d_meta = dev.upload(sub.meta)
            dev.map_async(d_data, sub.words, d_meta, sub.count, d_slice, out_offset_digests=b0 - lo)
            dev.sync()
            d_data.free()
            d_meta.free()
        height = cap_height if nslices > 1 else tree_height(hi - lo)
        d_scratch = dev.reduce_scratch(hi - lo, levels_variant)
        dev.reduce_async(d_slice, hi - lo, height, d_scratch, d_roots, root_index=s, levels_variant=levels_variant)
        dev.sync()
        d_scratch.free()
        d_slice.free()
    roots = dev.download(d_roots, 32 * nslices).reshape(-1, 8)
    d_roots.free()
    root = roots[0] if nslices == 1 else dev.combine(roots)
    return digest_hex(root)
