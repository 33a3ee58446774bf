#!/bin/bash
# tools/ab_libs.sh -- interleaved A/B of alternative builds of libvkmr_hip.so under bench.py (GPU box): bash tools/ab_libs.sh lib1.so lib2.so ...
cd ${GRAFT_REPO_ROOT:-.}
for round in 1 2; do for lib in default "$@"; do
  if [ "$lib" = default ]; then unset VKMR_HIP_LIB; else export VKMR_HIP_LIB=$lib; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-long-strings --no-clock-leg 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); v=d['valu_roofline']
print('$lib', 'ms/step', round(d['ms_per_step'],3), 'map', round(v['map_ms_per_step'],3), 'reduce', round(v['reduce_ms_per_step'],3), d['root_matches_golden'])"
done; done
