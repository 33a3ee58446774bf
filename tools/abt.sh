#!/bin/bash
# interleaved timing over VKMR_MAP_TILE values.  Usage: bash tools/abt.sh "<bench args>" 512 1024 ...
ARGS=$1; shift
# the knobs exist in the experiments build only (vk_merkle_roots_amd/build.py: build_experiments -> build/ab/libexp.so)
export VKMR_HIP_LIB=${VKMR_HIP_LIB:-${GRAFT_REPO_ROOT:-$(pwd)}/build/ab/libexp.so}
for round in 1 2; do
  for t in "$@"; do
    VKMR_MAP_TILE=$t timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --no-clock-leg --no-pipeline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
w = d['valu_roofline']
print('tile $t round', $round, 'ms/step', round(d['ms_per_step'], 3), 'map', round(w['map_ms_per_step'], 3), d['root'][:12])"
  done
done
