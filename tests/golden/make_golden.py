#!/usr/bin/env python3
"""Regenerates tests/golden/vectors.json from the REFERENCE ITSELF.

Runs only where /root/reference is present: oracle/Makefile compiles the reference's
own CPU-serial sources (unmodified, where they lie) into oracle/_ref/, and this
script records what that build prints/returns.  The fixture is data only: inputs
(or the seed that regenerates them) and the reference's outputs.

    python tests/golden/make_golden.py
"""
import ctypes as C
import hashlib
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")

LEAF_LENGTHS = [1, 2, 3, 4, 5, 31, 32, 54, 55, 56, 57, 62, 63, 64, 65, 118, 119, 120, 121, 126, 127, 128, 129,
                183, 184, 191, 192, 193, 247, 248, 255, 256, 1000, 4095, 4096, 4097]
TREE_COUNTS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 257, 1000, 1025]


def pattern(n, salt):
    """Deterministic bytes incl. high-bit values, no newline (so it can also go through stdin)."""
    out = bytearray()
    x = (salt * 2654435761 + 12345) & 0xFFFFFFFF
    while len(out) < n:
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        b = (x >> 24) & 0xFF
        if b == 0x0A:
            b = 0x8A
        out.append(b)
    return bytes(out)


def run_ref(stream):
    r = subprocess.run([os.path.join(REF, "vkmr_cpu_ref")], input=stream, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    line = [l for l in r.stdout.decode().splitlines() if "computed root" in l]
    if not line:
        return None
    text = line[0]
    root = text.split("=> ")[1].split(" in ")[0]
    items = int(text.split("(of ")[1].split(" item")[0])
    nbytes = int(text.split("item(s), ")[1].split(" byte")[0])
    return {"root": root, "items": items, "bytes": nbytes}


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all", "ref"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(REF, "libvkmr_ref.so"))
    vec = {"_about": "outputs of the reference CPU-serial path (g++ -O2 build of src/vkmr/{SHA-256plus,Debug,Inputs,StopWatch}.cpp); "
                     "regenerate with tests/golden/make_golden.py"}

    # per-leaf digests: reference vkmr::cpu_sha256d / cpu_sha256
    leaves = []
    for n in LEAF_LENGTHS:
        msg = pattern(n, n)
        d = C.create_string_buffer(32)
        s = C.create_string_buffer(32)
        lib.ref_sha256d(msg, C.c_size_t(n), d)
        lib.ref_sha256(msg, C.c_size_t(n), s)
        leaves.append({"len": n, "salt": n, "sha256": s.raw.hex(), "sha256d": d.raw.hex()})
    vec["leaves"] = leaves

    # tree roots: reference CpuSha256D over n strings of 1..40 bytes
    trees = []
    for n in TREE_COUNTS:
        strs = [pattern(1 + (i * 7) % 40, 1000 + i) for i in range(n)]
        blob = b"".join(strs)
        offs = (C.c_uint64 * (n + 1))()
        o = 0
        for i, s_ in enumerate(strs):
            offs[i] = o
            o += len(s_)
        offs[n] = o
        hexbuf = C.create_string_buffer(65)
        lib.ref_root(blob, offs, C.c_size_t(n), hexbuf)
        trees.append({"count": n, "root": hexbuf.value.decode()})
    vec["trees"] = trees

    # whole streams through the reference's stdin driver
    streams = {}
    alpha = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
    g1 = "".join("%02d%s\n" % (i, alpha) for i in range(16)).encode()
    streams["G1_strm16x64"] = dict(run_ref(g1), stream_sha256=hashlib.sha256(g1).hexdigest(), generator="strm", note="16 strings '%02d'+[a-zA-Z0-9], 64 B each")
    for name, args in (("G2_rndm_1712489279_1024_127", ["1712489279", "1024", "127"]),
                       ("G4_rndm_42_4096_4096", ["42", "4096", "4096"]),
                       ("G6_rndm_7_1000_300", ["7", "1000", "300"]),
                       ("G3_rndm_42_1048576_127", ["42", "1048576", "127"])):
        data = subprocess.run([os.path.join(REF, "rndm")] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        streams[name] = dict(run_ref(data), stream_sha256=hashlib.sha256(data).hexdigest(), generator="rndm " + " ".join(args))
    # line-rule cases: CR kept, empty lines skipped, no trailing newline, high-bit bytes
    cases = {
        "L1_crlf": b"alpha\r\nbeta\r\ngamma\r\n",
        "L2_empty_lines": b"\n\none\n\ntwo\n\n\nthree",
        "L3_no_trailing_newline": b"solo",
        "L4_single_with_newline": b"solo\n",
        "L5_highbit": bytes([0xFF, 0xFE, 0x80, 0x0A, 0x81, 0x0A, 0xC3, 0xA9, 0x0A]),
        "L6_two": b"a\nb\n",
        "L7_three": b"a\nb\nc\n",
    }
    for name, data in cases.items():
        streams[name] = dict(run_ref(data), stream_hex=data.hex())
    vec["streams"] = streams

    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(vec, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "vectors.json"))


if __name__ == "__main__":
    main()
