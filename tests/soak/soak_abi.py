"""Random streams through the C ABI with random slicings against the oracle (the oracle is the checker, so this lives
under tests/).  GPU box only:  python3 tests/soak/soak_abi.py [seconds]  (run from the repo root)."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, vk_merkle_roots_amd as vk
from conftest import Oracle
o = Oracle(); gpu = vk.HipDevice(0)
rng = np.random.default_rng(int(time.time()))
t0 = time.time(); cases = 0; proofs = 0; last = t0
while time.time() - t0 < (float(sys.argv[1]) if len(sys.argv) > 1 else 90):
    # random stream -> root via random slicing, both map modes via batched/unbatched engine paths
    n = int(rng.choice([rng.integers(1, 2000), rng.integers(1, 200000)]))
    maxlen = int(rng.choice([2, 20, 65, 127, 300, 2000]))
    seed = int(rng.integers(1, 2**31))
    b = vk.rndm_packed(seed, n, maxlen)
    want = o.hex(o.root(o.leaves_packed(b.data, b.meta, threads=16), threads=16))
    cap = 1 << int(rng.integers(1, 19))
    bs = int(rng.integers(1, max(2, n)))
    got1 = vk.merkle_root_packed(gpu, b, slice_capacity=cap, batch_strings=max(bs, n // 50 + 1))
    got2 = vk.merkle_root_packed_batched(gpu, b, slice_capacity=cap, batch_strings=max(bs, n // 50 + 1))
    got3 = vk.merkle_root_packed(gpu, b)
    assert got1 == want and got2 == want and got3 == want, (seed, n, maxlen, cap, bs)
    if cases % 3 == 0:   # proofs written by the reduction in its pass: same root, and every proof folds to it (host fold = the checker's rule)
        leaves = o.leaves_packed(b.data, b.meta, threads=16)
        height = vk.tree_height(n)
        k = int(rng.integers(1, 17))
        idx = [int(x) for x in rng.integers(0, n, size=k)]
        d_in = gpu.upload(leaves)
        sib, root = gpu.reduce_with_proofs(d_in, n, height, idx)
        assert o.hex(root) == want, ("proofs: root", seed, n, maxlen, idx)
        hl = vk.host_lib()
        for q, index in enumerate(idx):
            folded = np.zeros(8, dtype=np.uint32)
            hl.vkmr_host_cpu_fold_proof(leaves[index].ctypes.data, index, np.ascontiguousarray(sib[q]).ctypes.data, height, folded.ctypes.data)
            assert o.hex(folded) == want, ("proofs: fold", seed, n, maxlen, index)
        d_in.free()
        proofs += 1
    cases += 1
    if time.time() - last > 60:   # a progress line a minute (a silent GPU run is taken to be hung after seven)
        last = time.time(); print(f"... {cases} streams so far, all equal", flush=True)
print("soak ok:", cases, "random streams, all roots equal the oracle;", proofs, "of them also reduced with 1..16 proofs written in the pass, all folding to the root")
