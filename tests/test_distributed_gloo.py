"""The N>1 path on CPU: two gloo ranks, slices sharded across ranks, ONE gather of the
slice roots, combine on rank 0.  Hashing here is done by the product's own "CPU"
backend (csrc/host/cpu_sha256d.cpp through libvkmr_host.so) -- the GPU ranks of
bench.py run the same sharding/gather/combine code with the HIP kernels; the oracle
only checks the final root."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, seed, n, cap_log2, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import vk_merkle_roots_amd as vk
    from vk_merkle_roots_amd.distributed import gather_roots, shard_slices
    from vk_merkle_roots_amd.engine import digest_hex, tree_height
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h = vk.host_lib()
    cap = 1 << cap_log2
    total_slices = (n + cap - 1) // cap
    lo, hi = shard_slices(total_slices, world, rank)
    batch = vk.rndm_packed(seed, n, 127)          # every rank can regenerate the stream; it hashes only its own slices
    roots = np.zeros((hi - lo, 8), dtype=np.uint32)
    for k, s in enumerate(range(lo, hi)):
        sub = batch.slice(s * cap, min(n, (s + 1) * cap))
        leaves = np.zeros((sub.count, 8), dtype=np.uint32)
        h.vkmr_host_cpu_leaves(sub.data.ctypes.data, sub.meta.ctypes.data, sub.count, leaves.ctypes.data)
        height = cap_log2 if total_slices > 1 else tree_height(sub.count)
        assert h.vkmr_host_cpu_reduce(leaves.ctypes.data, sub.count, height, roots[k].ctypes.data) == 0
    allr = gather_roots(roots, dist, rank, world)
    if rank == 0:
        assert allr.shape == (total_slices, 8)
        top = np.zeros(8, dtype=np.uint32)
        if total_slices == 1:
            top = allr[0]
        else:
            assert h.vkmr_host_cpu_combine(np.ascontiguousarray(allr).ctypes.data, total_slices, top.ctypes.data) == 0
        with open(out_path, "w") as f:
            f.write(digest_hex(top))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,cap_log2", [(1024, 9), (5000, 10), (3000, 12), (4097, 10), (2048, 10)])
def test_two_ranks_sharded_slices(native, oracle, tmp_path, n, cap_log2):
    import torch.multiprocessing as mp
    import vk_merkle_roots_amd as vk
    out = str(tmp_path / "root.txt")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 77, n, cap_log2, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    b = vk.rndm_packed(77, n, 127)
    want = oracle.hex(oracle.root(oracle.leaves_packed(b.data, b.meta)))
    assert open(out).read() == want


@pytest.mark.parametrize("n,cap_log2", [(8 * 512, 9), (7 * 512 + 100, 9), (7 * 1024 + 1, 10)])
def test_eight_ranks_one_slice_each_ragged_last(native, oracle, tmp_path, n, cap_log2):
    """BASELINE configs[3]'s shape in small: eight ranks, ONE slice per rank, the last rank's slice short (it is
    still reduced to capacity height, reference Reductions.cpp:471), eight roots in one gather, combine on rank 0."""
    import torch.multiprocessing as mp
    import vk_merkle_roots_amd as vk
    out = str(tmp_path / "root.txt")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 8, port, 91, n, cap_log2, out)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    b = vk.rndm_packed(91, n, 127)
    want = oracle.hex(oracle.root(oracle.leaves_packed(b.data, b.meta)))
    assert open(out).read() == want


def test_shard_plan_covers_everything():
    from vk_merkle_roots_amd.distributed import shard_slices
    for total in range(0, 40):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_slices(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_cpu_backend_entry_points_match_oracle(native, oracle):
    import vk_merkle_roots_amd as vk
    h = vk.host_lib()
    b = vk.rndm_packed(5, 999, 300)
    leaves = np.zeros((b.count, 8), dtype=np.uint32)
    h.vkmr_host_cpu_leaves(b.data.ctypes.data, b.meta.ctypes.data, b.count, leaves.ctypes.data)
    assert (leaves == oracle.leaves_packed(b.data, b.meta)).all()
    for height in (10, 13):
        r = np.zeros(8, dtype=np.uint32)
        assert h.vkmr_host_cpu_reduce(leaves.ctypes.data, b.count, height, r.ctypes.data) == 0
        assert (r == oracle.reduce_height(leaves, height)).all()
    r = np.zeros(8, dtype=np.uint32)
    assert h.vkmr_host_cpu_combine(leaves.ctypes.data, 999, r.ctypes.data) == 0
    assert (r == oracle.root(leaves)).all()
