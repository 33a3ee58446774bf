#!/usr/bin/env python3
"""Turns a tools/pmc_profile.sh output directory into profiles/pmc_latest.json: HBM bytes per launch of the
map kernel and per step of the reduction, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KiB and
tallies the 128-byte requests of gfx950 as 64 bytes -> x2; WRITE_SIZE in KiB is exact).  The x2 holds for the
reduction's pair loads too: tools/fetch_calibrate.hip reads 1 GiB in that pattern and FETCH_SIZE x 2 returns
1.074e9 bytes, every TCC_EA0_RDREQ being a 128-byte request (profiles/r02_fetch_calibration.txt).

    python3 tools/pmc_to_json.py <pmc dir> <out.json> [--batch-log2 26] [--maxlen 127] [--slice-log2 26] [--launches-per-step 1]
"""
import argparse
import collections
import csv
import glob
import json
import os


def main():
    p = argparse.ArgumentParser()
    p.add_argument("root")
    p.add_argument("out")
    p.add_argument("--batch-log2", type=int, default=26)
    p.add_argument("--maxlen", type=int, default=127)
    p.add_argument("--slice-log2", type=int, default=26)
    p.add_argument("--launches-per-step", type=int, default=1)
    p.add_argument("--long-strings-dir", default=None,
                   help="FETCH_SIZE and WRITE_SIZE passes over tools/long_strings_probe.py (rndm 42 2^21 4096, one batch): adds the long_strings_* keys")
    a = p.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(a.root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
    build = None
    try:   # the identity of the kernel library that was profiled (vk_merkle_roots_amd/build.py: source_id)
        build = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vk_merkle_roots_amd", "libvkmr_hip.isa.json")))["build"]
    except (OSError, ValueError, KeyError):
        pass
    res = {"source": os.path.basename(a.root.rstrip("/")), "build": build, "strings_per_map_launch": 1 << a.batch_log2, "maxlen": a.maxlen,
           "slice_log2": a.slice_log2,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_profile.sh); read = FETCH_SIZE KiB x 1024 x 2 "
                     "(gfx950 tallies its 128-byte requests as 64 B; calibrated on 1 GiB for both read patterns, tools/fetch_calibrate.hip), "
                     "write = WRITE_SIZE KiB x 1024 (tools/pmc_to_json.py)"}
    steps = None
    red_read = red_write = 0.0
    for k, c in sorted(acc.items()):
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        short = k.split("<")[0]
        fetch = sum(c["FETCH_SIZE"]) * 1024 * 2
        write = sum(c["WRITE_SIZE"]) * 1024
        n = len(c["FETCH_SIZE"])
        res[short + "_hbm_read_bytes_per_launch"] = fetch / n
        res[short + "_hbm_write_bytes_per_launch"] = write / n
        res[short + "_hbm_bytes_per_launch"] = (fetch + write) / n
        res[short + "_launches_profiled"] = n
        if short == "map_kernel":
            res["map_kernel_variant"] = k
            res["map_kernel_symbol"] = k          # e.g. map_kernel<512, 1024, 17664, 0, false, 0>: matched against vkmr_hip_kernel_info()
            steps = n / a.launches_per_step
        if short.startswith("reduce_"):
            red_read += fetch
            red_write += write
        for extra in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"):
            if extra in c:
                res[short + "_" + extra + "_per_launch"] = sum(c[extra]) / len(c[extra])
    if steps:
        res["steps_profiled"] = steps
        res["reduce_hbm_read_bytes_per_step"] = red_read / steps
        res["reduce_hbm_write_bytes_per_step"] = red_write / steps
        res["reduce_hbm_bytes_per_step"] = (red_read + red_write) / steps
    if a.long_strings_dir:
        ls = collections.defaultdict(list)
        for path in glob.glob(os.path.join(a.long_strings_dir, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if "map_kernel" in r["Kernel_Name"]:
                    ls[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    res["long_strings_map_kernel_symbol"] = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if ls.get("FETCH_SIZE") and ls.get("WRITE_SIZE"):
            rd = sum(ls["FETCH_SIZE"]) / len(ls["FETCH_SIZE"]) * 1024 * 2
            wr = sum(ls["WRITE_SIZE"]) / len(ls["WRITE_SIZE"]) * 1024
            res["long_strings_workload"] = "rndm 42 2^21 4096, one batch"
            res["long_strings_map_hbm_read_bytes_per_launch"] = rd
            res["long_strings_map_hbm_write_bytes_per_launch"] = wr
            res["long_strings_map_hbm_bytes_per_launch"] = rd + wr
    json.dump(res, open(a.out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
