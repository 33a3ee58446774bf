"""The multi-GPU exchange of the C ABI on the one-GPU box: real RCCL with one rank (both ways of forming the
communicator), the slice-order kernel on hand-made "gathered" arrays, and the on-device combine.  More ranks
need more GPUs: N > 1 over real RCCL is UNMEASURED on this pool (the 8-GPU run is the driver's); the
same-process call sequence for several devices runs in tests/test_frontend.py through a stand-in for RCCL."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gather_once(gpu, comm, per_rank, roots):
    import vk_merkle_roots_amd as vk
    nr, nl = C.c_int(0), C.c_int(0)
    vk.check(gpu.lib.vkmr_hip_comm_size(comm, C.byref(nr), C.byref(nl)), "comm_size")
    assert (nr.value, nl.value) == (1, 1)
    d_mine = gpu.upload(roots)
    d_all = gpu.alloc(32 * per_rank)
    streams = (C.c_void_p * 1)(gpu.stream)
    mine = (C.c_void_p * 1)(d_mine.ptr)
    allp = (C.c_void_p * 1)(d_all.ptr)
    vk.check(gpu.lib.vkmr_hip_gather_roots_async(comm, streams, mine, per_rank, allp), "gather_roots")
    return gpu.download(d_all, 32 * per_rank).reshape(-1, 8)


def test_rccl_one_rank_init_all(gpu):
    import vk_merkle_roots_amd as vk
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    vk.check(gpu.lib.vkmr_hip_comm_init_all(devs, 1, C.byref(comm)), "comm_init_all")
    roots = np.random.default_rng(1).integers(0, 2**32, size=(3, 8), dtype=np.uint32)
    assert (_gather_once(gpu, comm, 3, roots) == roots).all()
    # which RCCL the communicator is bound to: a shared object that exists, and a version
    info = gpu.lib.vkmr_hip_comm_info().decode()
    assert info.startswith("rccl=/") and "librccl" in info and " version=" in info and int(info.rsplit("version=", 1)[1]) > 20000, info
    vk.check(gpu.lib.vkmr_hip_comm_destroy(comm), "comm_destroy")


def test_rccl_one_rank_by_unique_id(gpu):
    """What every rank of bench.py does: rank 0 makes the id, all join with it."""
    import vk_merkle_roots_amd as vk
    uid = C.create_string_buffer(vk._abi.COMM_ID_BYTES)
    vk.check(gpu.lib.vkmr_hip_comm_create_id(uid), "comm_create_id")
    comm = C.c_void_p()
    vk.check(gpu.lib.vkmr_hip_comm_init_rank(0, uid, 1, 0, C.byref(comm)), "comm_init_rank")
    roots = np.random.default_rng(2).integers(0, 2**32, size=(1, 8), dtype=np.uint32)
    assert (_gather_once(gpu, comm, 1, roots) == roots).all()
    vk.check(gpu.lib.vkmr_hip_comm_destroy(comm), "comm_destroy")


def test_comm_rejects_bad_arguments(gpu):
    import vk_merkle_roots_amd as vk
    comm = C.c_void_p()
    assert gpu.lib.vkmr_hip_comm_init_all(None, 1, C.byref(comm)) == vk._abi.ERR_INVALID
    uid = C.create_string_buffer(vk._abi.COMM_ID_BYTES)
    assert gpu.lib.vkmr_hip_comm_init_rank(0, uid, 2, 2, C.byref(comm)) == vk._abi.ERR_INVALID
    assert gpu.lib.vkmr_hip_gather_roots_async(None, None, None, 1, None) == vk._abi.ERR_INVALID


@pytest.mark.parametrize("nranks,total", [(1, 5), (2, 7), (3, 3), (4, 10), (8, 8), (8, 64), (8, 61), (5, 2), (7, 1000)])
def test_roots_in_slice_order(gpu, nranks, total):
    """Slice k (1-based) lives on rank (k-1) % nranks as that rank's entry (k-1) // nranks."""
    import vk_merkle_roots_amd as vk
    per = (total + nranks - 1) // nranks
    rng = np.random.default_rng(nranks * 1000 + total)
    want = rng.integers(0, 2**32, size=(total, 8), dtype=np.uint32)
    gathered = rng.integers(0, 2**32, size=(nranks * per, 8), dtype=np.uint32)      # padding cells hold garbage
    for k in range(total):
        gathered[(k % nranks) * per + k // nranks] = want[k]
    d_g, d_o = gpu.upload(gathered), gpu.alloc(32 * total)
    vk.check(gpu.lib.vkmr_hip_roots_in_slice_order_async(gpu.index, gpu.stream, d_g.ptr, nranks, per, total, d_o.ptr), "slice_order")
    assert (gpu.download(d_o, 32 * total).reshape(-1, 8) == want).all()
    assert gpu.lib.vkmr_hip_roots_in_slice_order_async(gpu.index, gpu.stream, d_g.ptr, nranks, per, nranks * per + 1, d_o.ptr) == vk._abi.ERR_INVALID


def test_combine_async_is_the_cpu_rule(gpu, oracle):
    """Device-resident roots, caller's stream, no allocation inside: duplicate-last tree, at least one level."""
    rng = np.random.default_rng(4)
    for n in [1, 2, 3, 8, 9, 100, 129, 5000]:
        roots = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32)
        d_in, d_root = gpu.upload(roots), gpu.alloc(32)
        d_scratch = gpu.reduce_scratch(n) if n > 128 else None
        gpu.combine_async(d_in, n, d_scratch, d_root)
        assert (gpu.download(d_root, 32) == oracle.root(roots)).all(), n
