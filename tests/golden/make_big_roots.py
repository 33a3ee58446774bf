#!/usr/bin/env python3
"""Generates tests/golden/big_roots.json: Merkle roots of the bench workloads at
BASELINE.json's full size, produced by the REFERENCE ITSELF.

For every seed s in 42..49 the stream `rndm s 67108864 127` (the reference's own rndm
tool, oracle/_ref/rndm, compiled from src/rndm/Rndm.cpp) is written to a scratch file and fed to the reference's
own CPU-serial backend (oracle/_ref/vkmr_cpu_ref: src/vkmr/{SHA-256plus,Inputs,...}.cpp
behind the run() loop of oracle/ref_driver.cpp), and the root, item and byte counts it
prints are recorded.  One such run takes about five minutes and 7.5 GB of memory, so
they go `--jobs` at a time (default 2).  Runs only where /root/reference is present.

`combined[N]` is what bench.py --gpus N must print: rank r holds the 2^26 leaves of seed
42+r as ONE slice, the N slice roots are combined in rank (= slice) order with the
duplicate-last rule (reference src/vkmr/Reductions.cpp:703-712); for N == 1 the slice root
is the root (Reductions.cpp:692-701).  The combine is computed here twice -- with the pinned
oracle (oracle/liboracle.so) and with hashlib -- and must agree.

    python tests/golden/make_big_roots.py [--jobs 2] [--seeds 42 43 ...] [--count-log2 26]
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(HERE, "big_roots.json")


def run_one(seed, count, maxlen):
    t0 = time.time()
    # through a scratch file, not a pipe: rndm flushes every line (src/rndm/Rndm.cpp:66) and a pipe
    # turns that into one context switch per 64-byte line (3x slower than the two programs back to back)
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"vkmr_big_{seed}_{count}_{maxlen}.txt")
    with open(tmp, "wb") as f:
        subprocess.check_call([os.path.join(REF, "rndm"), str(seed), str(count), str(maxlen)], stdout=f, stderr=subprocess.DEVNULL)
    try:
        with open(tmp, "rb") as f:
            out = subprocess.run([os.path.join(REF, "vkmr_cpu_ref")], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    finally:
        os.unlink(tmp)
    line = [l for l in out.splitlines() if "computed root" in l][0]
    rec = {"root": line.split("=> ")[1].split(" in ")[0],
           "items": int(line.split("(of ")[1].split(" item")[0]),
           "bytes": int(line.split("item(s), ")[1].split(" byte")[0]),
           "reference_ms": float(line.rsplit(" in ", 1)[1])}
    sys.stderr.write(f"seed {seed}: {rec['root']} ({rec['items']} items, {time.time() - t0:.0f} s)\n")
    return seed, rec


def d(b):
    return hashlib.sha256(hashlib.sha256(b).digest()).digest()


def combine_hashlib(roots_hex):
    nodes = [bytes.fromhex(h) for h in roots_hex]
    if len(nodes) == 1:
        return nodes[0].hex()
    while True:
        if len(nodes) & 1:
            nodes.append(nodes[-1])
        nodes = [d(nodes[i] + nodes[i + 1]) for i in range(0, len(nodes), 2)]
        if len(nodes) == 1:
            return nodes[0].hex()


def combine_oracle(roots_hex):
    import numpy as np
    if len(roots_hex) == 1:
        return roots_hex[0]
    L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    L.oracle_root_inplace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.oracle_words_to_hex.argtypes = [C.c_void_p, C.c_char_p]
    work = np.array([np.frombuffer(bytes.fromhex(h), dtype=">u4").astype(np.uint32) for h in roots_hex], dtype=np.uint32)
    out = np.zeros(8, dtype=np.uint32)
    assert L.oracle_root_inplace(work.ctypes.data, len(roots_hex), out.ctypes.data) == 0
    buf = C.create_string_buffer(65)
    L.oracle_words_to_hex(out.ctypes.data, buf)
    return buf.value.decode()


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--jobs", type=int, default=2)
    p.add_argument("--seeds", type=int, nargs="*", default=list(range(42, 50)))
    p.add_argument("--count-log2", type=int, default=26)
    p.add_argument("--maxlen", type=int, default=127)
    p.add_argument("--config5", action="store_true",
                   help="only add rec['config5']: the reference's root of `rndm 42 2^24 4096` (BASELINE configs[4]; 34 GB of text, about 40 minutes)")
    a = p.parse_args()
    if a.config5:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all", "ref"], stdout=subprocess.DEVNULL)
        rec = json.load(open(OUT))
        _, r = run_one(42, 1 << 24, 4096)
        rec["config5"] = dict(r, generator="rndm 42 16777216 4096",
                              about="BASELINE configs[4] at full size, printed by the reference CPU-serial path; bench.py's config5_full leg checks its root against this")
        with open(OUT, "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)
        return
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all", "ref"], stdout=subprocess.DEVNULL)
    count = 1 << a.count_log2
    rec = {}
    if os.path.exists(OUT):
        rec = json.load(open(OUT))
        if rec.get("count") != count or rec.get("maxlen") != a.maxlen:
            rec = {}
    rec.update({"_about": "roots printed by the reference CPU-serial path (oracle/_ref/rndm | oracle/_ref/vkmr_cpu_ref, g++ -O2) for "
                          "`rndm <seed> <count> <maxlen>`; combined[N] = duplicate-last tree over the roots of seeds 42..42+N-1 in order "
                          "(N = 1: the root itself); regenerate with tests/golden/make_big_roots.py",
                "count": count, "maxlen": a.maxlen})
    subs = rec.setdefault("sub_roots", {})
    todo = [s for s in a.seeds if str(s) not in subs]
    with ThreadPoolExecutor(max_workers=max(1, a.jobs)) as ex:
        for seed, r in ex.map(lambda s: run_one(s, count, a.maxlen), todo):
            subs[str(seed)] = r
            with open(OUT, "w") as f:   # keep what is done if the run is interrupted
                json.dump(rec, f, indent=1, sort_keys=True)
    comb = {}
    for n in (1, 2, 3, 4, 5, 6, 7, 8):
        if all(str(42 + r) in subs for r in range(n)):
            roots = [subs[str(42 + r)]["root"] for r in range(n)]
            h = combine_hashlib(roots)
            assert h == combine_oracle(roots), n
            comb[str(n)] = h
    rec["combined"] = comb
    with open(OUT, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
