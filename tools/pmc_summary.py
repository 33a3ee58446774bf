#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import collections
import csv
import glob
import os
import sys


def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(acc):
        d = dur.get(k, [])
        print(f"== {k}: dispatches/pass~{len(d) // max(1, len(glob.glob(os.path.join(root, '*.log'))))} mean_us={sum(d) / max(1, len(d)):.1f}")
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"   {c:28s} mean={sum(v) / len(v):.4g}  n={len(v)}")
        a = {c: sum(v) / len(v) for c, v in acc[k].items()}
        if "FETCH_SIZE" in a:
            print(f"   -> HBM read bytes/launch  = {a['FETCH_SIZE'] * 1024 * 2:.4g} (FETCH_SIZE KiB x 2: gfx950 counts 128-B requests as 64 B)")
        if "WRITE_SIZE" in a:
            print(f"   -> HBM write bytes/launch = {a['WRITE_SIZE'] * 1024:.4g}")
        if "SQ_ACTIVE_INST_VALU" in a and "SQ_BUSY_CYCLES" in a:
            pass


if __name__ == "__main__":
    main(sys.argv[1])
