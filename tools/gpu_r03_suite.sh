# round 3: GPU suite, smoke, then the first full bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r03/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/r03/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc=$?"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_full.json 2> gpurun_out/r03/bench_full.err; echo "bench rc=$?"; tail -3 gpurun_out/r03/bench_full.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r03/bench_full.json').read().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','root_matches_golden']}); print(d['roofline']); print(d['roofline_reduce'])
v=d['valu_roofline']; print(v.get('clock')); print(v.get('map')); print(v.get('reduce'))
for k in ('reference_shapes','two_stream_overlap','pipeline_pcie_inclusive','cpu_baseline'): print(k, d.get(k))
print(d['long_strings']['map_ms'], d['long_strings']['roofline'])"
