cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-long-strings 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); print('one stream', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e9,3), 'G/s | two streams', d['two_stream_overlap'])"; done
