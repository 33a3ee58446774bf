#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace --memory-copy-trace run of `vkmr hip:0 < file`: how much of the host-to-device copy time
runs while a map kernel is executing (the front end issues the copies of batch k+1 on a copy stream, the map kernel of batch k
on the map stream: csrc/host/mappings.cpp).  Usage: python3 tools/overlap_from_trace.py <rocprofv3 output dir>"""
import csv
import glob
import os
import sys


def intervals(path, name_col, want):
    out = []
    for r in csv.DictReader(open(path)):
        if want(r.get(name_col, "")):
            out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return sorted(out)


def main(root):
    kt = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    mt = glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True)
    if not kt or not mt:
        sys.exit(f"no kernel / memory-copy trace under {root}")
    maps = intervals(kt[0], "Kernel_Name", lambda n: "map_kernel" in n)
    copies = intervals(mt[0], "Direction", lambda d: "HOST_TO_DEVICE" in d.upper() or "H2D" in d.upper())
    if not copies:
        copies = intervals(mt[0], "Direction", lambda d: True)
    tot = sum(e - s for s, e in copies)
    ov = 0
    for cs, ce in copies:
        for ms, me in maps:
            if me <= cs:
                continue
            if ms >= ce:
                break
            ov += min(ce, me) - max(cs, ms)
    span = (max(e for _, e in maps + copies) - min(s for s, _ in maps + copies)) if maps and copies else 0
    print(f"{len(maps)} map launches ({sum(e - s for s, e in maps) / 1e6:.2f} ms), {len(copies)} copies ({tot / 1e6:.2f} ms) over {span / 1e6:.2f} ms; "
          f"{ov / 1e6:.2f} ms of copy time ({100.0 * ov / max(tot, 1):.0f} %) ran while a map kernel was executing")


if __name__ == "__main__":
    main(sys.argv[1])
