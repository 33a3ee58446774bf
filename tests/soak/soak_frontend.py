"""Random streams through the C++ front end (`vkmr hip:0`, and `vkmr hip:all` over an aliased GPU) with random
pipeline shapes -- slice size, batch size, mappings in flight, slice budget -- against the oracle.  GPU box only.
    python3 tests/soak/soak_frontend.py [seconds]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import Oracle, build_virt_devices  # noqa: E402

o = Oracle()
vkmr = os.path.join(ROOT, "vk_merkle_roots_amd", "bin", "vkmr")
rndm = os.path.join(ROOT, "vk_merkle_roots_amd", "bin", "rndm")
virt = build_virt_devices()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = np.random.default_rng(int(time.time()))
t0 = time.time()
last = t0
cases = proofs = files = 0
while time.time() - t0 < budget:
    n = int(rng.choice([rng.integers(1, 3000), rng.integers(1, 300000)]))
    maxlen = int(rng.choice([1, 2, 5, 20, 65, 127, 129, 300, 3000]))
    seed = int(rng.integers(1, 2**31))
    stream = subprocess.run([rndm, str(seed), str(n), str(maxlen)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    want, cnt, nb = o.root_of_stream(stream)
    env = dict(os.environ, VKMR_SLICE_LOG2=str(int(rng.integers(1, 19))), VKMR_BATCH_BYTES=str(int(rng.integers(4096, 1 << 22))),
               VKMR_MAX_INFLIGHT=str(int(rng.integers(1, 6))), VKMR_PACK_THREADS=str(int(rng.integers(1, 17))))
    if rng.integers(0, 4) == 0:
        del env["VKMR_BATCH_BYTES"]      # the default batches (one span of stdin each)
    if rng.integers(0, 2):
        env["VKMR_INPUT_SPAN_MB"] = str(int(rng.choice([1, 2, 5, 32])))
    if rng.integers(0, 3) == 0:
        env["VKMR_DEVICE_SPLIT"] = "1"     # large spans are split into strings on the device
    if rng.integers(0, 2):
        env["VKMR_SLICE_BUDGET"] = str(int(rng.integers(1, 4)))
    backend = "hip:0"
    if rng.integers(0, 3) == 0:
        env.update(LD_PRELOAD=virt, VKMR_TEST_VIRTUAL_DEVICES=str(int(rng.integers(2, 9))))
        backend = "hip:all"
    proof_index = int(rng.integers(0, cnt)) if cnt and rng.integers(0, 3) == 0 else None
    if proof_index is not None:
        env["VKMR_PROOF_INDEX"] = str(proof_index)
    from_file = bool(rng.integers(0, 2))   # a regular file is mapped and handed out in spans; a pipe is read by Input's thread
    if from_file:
        path = f"/tmp/soak_frontend_{os.getpid()}.txt"
        with open(path, "wb") as f:
            f.write(stream)
        with open(path, "rb") as f:
            r = subprocess.run([vkmr, backend], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
        os.unlink(path)
        files += 1
    else:
        r = subprocess.run([vkmr, backend], input=stream, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    line = [l for l in r.stdout.decode().splitlines() if "computed root" in l]
    shape = {k: v for k, v in env.items() if k.startswith("VKMR_")}
    if cnt == 0:   # nothing but empty lines (rndm * * 1 writes none at all): no root line (reference run(), Vkmr.cpp:52)
        assert r.returncode == 0 and not line, (seed, n, maxlen, backend, from_file, r.stdout[-300:])
        cases += 1
        continue
    assert r.returncode == 0 and line, (seed, n, maxlen, backend, from_file, shape, r.stderr[-300:])
    assert f"(of {cnt} item(s), {nb} byte(s)) => {want} in" in line[-1], (seed, n, maxlen, backend, from_file, shape, line[-1])
    if proof_index is not None:   # the printed Merkle proof folds to the printed root
        import hashlib
        d = lambda b: hashlib.sha256(hashlib.sha256(b).digest()).digest()
        pl = [l for l in r.stdout.decode().splitlines() if l.startswith("proof: ")]
        assert pl and pl[0].split()[2] == str(proof_index), (seed, n, proof_index, pl[:1])
        cur = bytes.fromhex(pl[0].split()[-1])
        for l in pl[1:]:
            side, sib = l.split()[3], bytes.fromhex(l.split()[4])
            cur = d(sib + cur) if side == "sibling-on-left" else d(cur + sib)
        assert cur.hex() == want, (seed, n, maxlen, backend, shape, proof_index)
        proofs += 1
    cases += 1
    if time.time() - last > 60:   # a progress line a minute (a silent GPU run is taken to be hung after seven)
        last = time.time()
        print(f"... {cases} streams so far, all equal", flush=True)
print("front-end soak ok:", cases, "random streams and pipeline shapes (", files, "from a mapped file, the others through a pipe ), all roots equal the oracle;", proofs,
      "random Merkle proofs fold to their root")
