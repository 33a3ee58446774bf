#!/bin/bash
# FETCH_SIZE / time of the long-string map launch (rndm 42 2^21 4096, one batch), per fetch mode (GPU box).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export VKMR_HIP_LIB=${VKMR_HIP_LIB:-$REPO/build/ab/libexp.so}   # the knobs exist in the experiments build only
cd /tmp && export TMPDIR=/tmp
for v in 0 3; do
  rm -rf /tmp/fc_$v
  VKMR_MAP_VARIANT=$v timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fc_$v -- python3 $REPO/bench.py --leaves-log2 21 --maxlen 4096 --batch-log2 21 --slice-log2 21 --steps 2 --warmup 1 --no-cpu-baseline --no-clock-leg --no-pipeline > /tmp/fc_$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
t = sys.argv[1]
f = [float(r["Counter_Value"]) for p in glob.glob(f"/tmp/fc_{t}/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(p)) if "map_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for p in glob.glob(f"/tmp/fc_{t}/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(p)) if "map_kernel" in r["Kernel_Name"]]
print(f"variant {t}: FETCH_SIZE raw {sum(f)/len(f)*1024/1e6:.1f} MB per launch (packed input 4314 MB incl. metadata), kernel {sum(d)/len(d):.1f} us")
PY
done
