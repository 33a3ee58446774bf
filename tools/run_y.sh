cd $GRAFT_REPO_ROOT
for cfg in "0 0" "8 0" "7 99" "0 0" "8 0"; do set -- $cfg; VKMR_MAP_VARIANT=$1 VKMR_MAP_FIT=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-long-strings 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); v=d['valu_roofline']
print('variant $1 fit $2', 'ms/step', round(d['ms_per_step'],3), 'map', round(v['map_ms_per_step'],3), 'T/s', round(v['map_achieved_tops'],2), d['root_matches_golden'])"; done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -q -x > gpurun_out/pytest_map_new_default.log 2>&1; tail -2 gpurun_out/pytest_map_new_default.log
