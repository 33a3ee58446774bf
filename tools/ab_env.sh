#!/bin/bash
# tools/ab_env.sh -- interleaved A/B under bench.py in one gpurun call (GPU box).  Each argument is "label:VAR=value,VAR=value"
# (VKMR_HIP_LIB, VKMR_MAP_VARIANT ...); "default:" is the product library with no knob.
cd ${GRAFT_REPO_ROOT:-.}
BENCH_ARGS=${BENCH_ARGS:---steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-long-strings --no-clock-leg}
for round in 1 2; do for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  ( OLDIFS=$IFS; IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; IFS=$OLDIFS
    timeout -k 10 200 python bench.py $BENCH_ARGS 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); v=d['valu_roofline']
print('$label', 'ms/step', round(d['ms_per_step'],3), 'map', round(v['map_ms_per_step'],3), 'reduce', round(v['reduce_ms_per_step'],3), d['root_matches_golden'])" )
done; done
