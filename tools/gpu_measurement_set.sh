# final PMC set of round 2 (GPU box): bench passes + long-string passes -> profiles/pmc_latest.json; kernel stats; bench line
cd $GRAFT_REPO_ROOT
bash tools/pmc_profile.sh r02c > gpurun_out/pmc_r02c.log 2>&1
( cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r02c_long/$c -- python3 $GRAFT_REPO_ROOT/tools/long_strings_probe.py > /dev/null 2>&1; done )
python3 tools/pmc_to_json.py gpurun_out/pmc_r02c gpurun_out/pmc_r02c.json --long-strings-dir gpurun_out/pmc_r02c_long > /dev/null
cp gpurun_out/pmc_r02c.json profiles/pmc_latest.json
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline > $GRAFT_REPO_ROOT/gpurun_out/prof_r02c.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r02c.err )
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_e.json 2> gpurun_out/bench_r02_e.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_r02_e.json').read().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','root_matches_golden']}); print(d['roofline']); print(d['roofline_reduce']); print(d['long_strings']['roofline'])"
