#!/bin/bash
# interleaved timing over VKMR_MAP_TILE values.  Usage: bash tools/abt.sh "<bench args>" 512 1024 ...
ARGS=$1; shift
for round in 1 2; do
  for t in "$@"; do
    VKMR_MAP_TILE=$t timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --no-pipeline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
w = d['valu_roofline']
print('tile $t round', $round, 'ms/step', round(d['ms_per_step'], 3), 'map', round(w['map_ms_per_step'], 3), 'map_tops', round(w['map_achieved_tops'],1), d['root'][:12])"
  done
done
