"""Sharding plan of the Merkle tree over ranks, and the HOST-SIDE gather used where RCCL cannot
run: the world-2 / world-8 gloo tests on CPU and `bench.py --rehearse-gloo`.

On GPUs the roots travel through the C ABI instead (vkmr_hip_gather_roots_async: one RCCL
all-gather over xGMI, include/vkmr_hip.h); this module is the CPU-testable twin of that step
with the same sharding and ordering rules.

The reference drives a single device and combines slice roots on the host
(src/vkmr/Reductions.cpp:703-712); slices are independent sub-trees by design
(README.md:94-96), so they shard with no data-path collective.  The only exchange is
32 bytes per slice: `gather_roots` moves them to rank 0 with ONE collective
(a `torch.distributed` gather) in global slice order, which is what the combine needs.

Layout: the stream's slices are numbered 0..S-1; rank r owns the contiguous block
[r*S/W, (r+1)*S/W) ("slices shard one-per-GPU" when S == W).  Every slice but the
globally last is full, and every slice is reduced to log2(capacity) levels when S > 1.
"""
import numpy as np


def shard_slices(total_slices, world, rank):
    """Contiguous block of slice indices owned by `rank` (balanced to within one)."""
    base, extra = divmod(total_slices, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_roots(local_roots, dist, rank, world, device=None, equal_counts=False):
    """One gather of every rank's slice roots to rank 0.

    local_roots: uint32 array [k, 8].  k may differ per rank by at most one (ranks pad
    to the common maximum so a single fixed-size gather suffices; `equal_counts` skips
    the size agreement when the caller knows every rank holds the same k).  Returns
    the [S, 8] uint32 array in global slice order on rank 0, None elsewhere.
    """
    import torch
    local_roots = np.ascontiguousarray(local_roots, dtype=np.uint32).reshape(-1, 8)
    if world == 1:
        return local_roots
    k = local_roots.shape[0]
    kmax = k
    if not equal_counts:
        t = torch.tensor([k], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kmax = int(t.item())
    padded = np.zeros((kmax, 9), dtype=np.int64)          # last column: 1 = real root
    padded[:k, :8] = local_roots.astype(np.int64)
    padded[:k, 8] = 1
    mine = torch.from_numpy(padded)
    if device is not None:
        mine = mine.to(device)
    bucket = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, bucket, dst=0)
    if rank != 0:
        return None
    rows = torch.cat(bucket).cpu().numpy()
    rows = rows[rows[:, 8] == 1]
    return rows[:, :8].astype(np.uint32)
