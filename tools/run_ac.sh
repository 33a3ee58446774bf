cd $GRAFT_REPO_ROOT
for ml in 4096 600; do for v in 0 4 0 4; do VKMR_MAP_VARIANT=$v python3 tools/long_strings_probe.py 21 $ml; done; done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); print(d['long_strings'])"
