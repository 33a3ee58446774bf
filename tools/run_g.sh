cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()"
for lds in 0 20000 45000 60000 110000; do VKMR_MAP_DYNLDS=$lds python3 tools/long_strings_probe.py; done
for t in 256 512 1024 2048; do VKMR_MAP_TILE=$t python3 tools/long_strings_probe.py; done
cd /tmp && export TMPDIR=/tmp
for lds in 0 60000; do
  VKMR_MAP_DYNLDS=$lds timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ls_fetch_$lds -- python3 $GRAFT_REPO_ROOT/tools/long_strings_probe.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
rows=[r for p in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/ls_fetch_$lds/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(p)) if "map_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
v=[float(r["Counter_Value"]) for r in rows]
print("dynlds $lds: FETCH_SIZE x2 per launch = %.3f GB (algorithmic 4.31 GB), n=%d" % (sum(v)/len(v)*1024*2/1e9, len(v)))
PY
done
