cd $GRAFT_REPO_ROOT
VKMR_MAP_VARIANT=5 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_fullsize.py -q -x -k "map or config5_long or random" > gpurun_out/pytest_variant5.log 2>&1; tail -5 gpurun_out/pytest_variant5.log
for v in 0 5 0 5; do VKMR_MAP_VARIANT=$v python3 tools/long_strings_probe.py; done
cd /tmp && export TMPDIR=/tmp
for v in 5; do
  VKMR_MAP_VARIANT=$v timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ls_fetch_v$v -- python3 $GRAFT_REPO_ROOT/tools/long_strings_probe.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
rows=[r for p in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/ls_fetch_v$v/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(p)) if "map_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
v=[float(r["Counter_Value"]) for r in rows]
print("variant $v: FETCH_SIZE x2 per launch = %.3f GB (algorithmic 4.31 GB), n=%d" % (sum(v)/len(v)*1024*2/1e9, len(v)))
PY
done
