#!/bin/bash
# HBM read traffic (FETCH_SIZE) and time of the map kernel in per-lane fetch mode (VKMR_MAP_VARIANT=4)
# as a function of the tile size (GPU box).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export VKMR_HIP_LIB=${VKMR_HIP_LIB:-$REPO/build/ab/libexp.so}   # the knobs exist in the experiments build only
cd /tmp && export TMPDIR=/tmp
for t in 256 512 1024 2048; do
  rm -rf /tmp/ft_$t
  VKMR_MAP_VARIANT=4 VKMR_MAP_TILE=$t timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/ft_$t -- python3 $REPO/bench.py --leaves-log2 25 --steps 2 --warmup 1 --no-cpu-baseline --no-clock-leg > /tmp/ft_$t.log 2>&1
  python3 - $t <<'PY'
import csv, glob, sys
t = sys.argv[1]
f = [float(r["Counter_Value"]) for p in glob.glob(f"/tmp/ft_{t}/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(p)) if r["Kernel_Name"].startswith("void map_kernel") and r["Counter_Name"] == "FETCH_SIZE"]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for p in glob.glob(f"/tmp/ft_{t}/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(p)) if r["Kernel_Name"].startswith("void map_kernel")]
print(f"tile {t}: FETCH_SIZE raw {sum(f)/len(f)*1024/1e6:.1f} MB per launch of 2^23 strings (algorithmic read 612 MB), kernel {sum(d)/len(d):.1f} us")
PY
done
