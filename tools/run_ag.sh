cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_f.json 2> gpurun_out/bench_r02_f.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_r02_f.json').read().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','root_matches_golden']}); print(d['long_strings']['map_ms'], d['long_strings']['roofline']['frac'], d['two_stream_overlap']['ms_per_step'], d['cpu_baseline']['value'])"
