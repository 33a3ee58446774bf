cd $GRAFT_REPO_ROOT
for lg in 15 17 19 21; do python3 tools/long_strings_probe.py $lg 127; VKMR_HIP_LIB=build/ab/libold.so python3 tools/long_strings_probe.py $lg 127; done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -q -x > gpurun_out/pytest_spread.log 2>&1; tail -2 gpurun_out/pytest_spread.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-long-strings 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), d['root_matches_golden'])"
