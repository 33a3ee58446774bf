cd $GRAFT_REPO_ROOT
for f in 95 100 95 100; do VKMR_MAP_FIT=$f timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-long-strings 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); v=d['valu_roofline']
print('fit $f', 'ms/step', round(d['ms_per_step'],3), 'map', round(v['map_ms_per_step'],3), 'T/s', round(v['map_achieved_tops'],2), d['root_matches_golden'])"; done
