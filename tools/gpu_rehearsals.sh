cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 tests/soak/soak_abi.py 75 > gpurun_out/soak_r02.txt 2>&1; tail -2 gpurun_out/soak_r02.txt
timeout -k 10 300 python3 tests/soak/soak_frontend.py 120 > gpurun_out/soak_frontend_r02.txt 2>&1; tail -3 gpurun_out/soak_frontend_r02.txt
# the driver's launch form, rehearsed with two ranks on the one GPU (roots over gloo, labelled as such)
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse-gloo --steps 3 --warmup 1 > gpurun_out/bench_r02_torchrun_gloo2.json 2> gpurun_out/bench_r02_torchrun_gloo2.err; echo "torchrun rehearsal rc=$?"; head -c 400 gpurun_out/bench_r02_torchrun_gloo2.json; echo
# the same without the rehearsal flag must FAIL on a one-GPU box (rank 1 has no GPU), and print no result line
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 2 > gpurun_out/bench_r02_torchrun_2.json 2> gpurun_out/bench_r02_torchrun_2.err; echo "torchrun N=2 on one GPU rc=$? (must be non-zero)"; wc -c gpurun_out/bench_r02_torchrun_2.json; grep -h "LOCAL_RANK 1 but" gpurun_out/bench_r02_torchrun_2.err | head -2
