# PMC passes + kernel stats + bench on the current build (GPU box)
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/pytest_parity.log 2>&1; tail -3 gpurun_out/pytest_parity.log
bash tools/pmc_profile.sh r02a > gpurun_out/pmc_r02a.log 2>&1
python3 tools/pmc_to_json.py gpurun_out/pmc_r02a gpurun_out/pmc_r02a.json > /dev/null
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline --no-long-strings > $GRAFT_REPO_ROOT/gpurun_out/prof_r02a.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r02a.err )
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_b.json 2> gpurun_out/bench_r02_b.err; echo "bench rc=$?"
