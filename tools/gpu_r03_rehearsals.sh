# round 3: soaks against the oracle on the final kernels, and the driver's launch form rehearsed with two ranks on the one GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 300 python3 tests/soak/soak_abi.py 150 > gpurun_out/r03/soak_abi.txt 2>&1; tail -2 gpurun_out/r03/soak_abi.txt
timeout -k 10 400 python3 tests/soak/soak_frontend.py 200 > gpurun_out/r03/soak_frontend.txt 2>&1; tail -3 gpurun_out/r03/soak_frontend.txt
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse-gloo --steps 3 --warmup 1 > gpurun_out/r03/bench_torchrun_gloo2.json 2> gpurun_out/r03/bench_torchrun_gloo2.err; echo "torchrun rehearsal rc=$?"; python3 -c "
import json
d=json.loads(open('gpurun_out/r03/bench_torchrun_gloo2.json').read().splitlines()[-1]); print({k:d.get(k) for k in ('n_gpus','ms_per_step','root_matches_golden','sub_roots_match_golden')}, d['config']['collective'])"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 2 > gpurun_out/r03/bench_torchrun_2.json 2> gpurun_out/r03/bench_torchrun_2.err; echo "torchrun N=2 on one GPU rc=$? (must be non-zero)"; wc -c gpurun_out/r03/bench_torchrun_2.json
timeout -k 10 200 python bench.py --force-dist --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline --no-long-strings --no-clock-leg > gpurun_out/r03/bench_force_dist.json 2> gpurun_out/r03/bench_force_dist.err; echo "force-dist rc=$?"; python3 -c "
import json
d=json.loads(open('gpurun_out/r03/bench_force_dist.json').read().splitlines()[-1]); print(d['config']['rccl'], d['gather_ms'], d['root_matches_golden'])"
