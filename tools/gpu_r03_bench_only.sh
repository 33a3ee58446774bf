cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r03/bench.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r03/bench.json').read().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','root_matches_golden']}); print(d['roofline']['frac'], d['roofline']['traffic_source']['used'])
for k in ('from_file_on_stdin','pipeline_pcie_inclusive','two_stream_overlap'): print(k, d.get(k))"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "warm_up or golden" 2>&1 | tail -3
