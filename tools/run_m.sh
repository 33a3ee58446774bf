cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_r02_d.log 2>&1; tail -4 gpurun_out/pytest_r02_d.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_d.json 2> gpurun_out/bench_r02_d.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open("gpurun_out/bench_r02_d.json").read().splitlines()[-1])
print({k:d.get(k) for k in ["value","ms_per_step","root_matches_golden"]}); print(d["long_strings"])
PY
vk_merkle_roots_amd/bin/rndm 42 4194304 4096 > /tmp/l22.txt 2>/dev/null
for i in 1 2; do vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/l22.txt 2>/dev/null | tail -1; done
