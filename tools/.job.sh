cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
E=$PWD/build/ab/libexp.so
export VKMR_HIP_LIB=$E
{
for shape in "22 300" "22 700" "21 1500" "20 16000"; do set -- $shape
  for v in 0 24; do VKMR_MAP_VARIANT=$v timeout -k 10 200 python3 tools/long_strings_probe.py $1 $2; done
done
} > gpurun_out/r04/medium_two_blocks.txt 2>&1
( cd /tmp && export TMPDIR=/tmp; for shape in "22 300" "22 700" "21 1500"; do set -- $shape; for v in 0 24; do export VKMR_MAP_VARIANT=$v; timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r04_two/m$2_v$v -- python3 $GRAFT_REPO_ROOT/tools/long_strings_probe.py $1 $2 > /dev/null 2>&1; done; done )
python3 - <<'PY' >> gpurun_out/r04/medium_two_blocks.txt
import csv, glob, collections
for m in (300, 700, 1500):
  for v in (0, 24):
    acc = collections.defaultdict(list)
    for path in glob.glob(f"gpurun_out/pmc_r04_two/m{m}_v{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == "FETCH_SIZE":
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, vals in acc.items():
        if "map_kernel" in k:
            print(f"maxlen {m} variant {v}: {k}: FETCH_SIZE x 2 KiB = {sum(vals) * 2048 / len(vals) / 1e9:.3f} GB per launch ({len(vals)} launches)")
PY
cat gpurun_out/r04/medium_two_blocks.txt
