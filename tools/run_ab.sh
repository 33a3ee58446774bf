cd $GRAFT_REPO_ROOT
for w in 5 40 120 5 120; do timeout -k 10 200 python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-pipeline --no-long-strings 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); v=d['valu_roofline']
print('warmup $w', 'ms/step', round(d['ms_per_step'],3), 'map', round(v['map_ms_per_step'],3), 'reduce', round(v['reduce_ms_per_step'],3))"; done
for s in 20 100 400; do timeout -k 10 200 python bench.py --steps $s --warmup 5 --no-cpu-baseline --no-pipeline --no-long-strings 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); v=d['valu_roofline']
print('steps $s warmup 5', 'ms/step', round(d['ms_per_step'],3), 'map', round(v['map_ms_per_step'],3), 'reduce', round(v['reduce_ms_per_step'],3))"; done
