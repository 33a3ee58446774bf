#!/bin/bash
# tools/ab.sh -- interleaved A/B timing of alternative builds of libvkmr_hip.so in ONE gpurun call
# (same device, same process conditions).  Usage: bash tools/ab.sh "<bench args>" libA.so libB.so ...
ARGS=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    VKMR_HIP_LIB=$lib timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
v = d['valu_roofline']
print('$lib', 'round', $round, 'ms/step', round(d['ms_per_step'], 3), 'map', round(v['map_ms_per_step'], 3), 'reduce', round(v['reduce_ms_per_step'], 3), d['root'][:12])"
  done
done
