cd $GRAFT_REPO_ROOT
# two ranks forced onto the one GPU through HIP_VISIBLE_DEVICES: RCCL must refuse (duplicate GPU) and the run must fail quickly, not hang
HIP_VISIBLE_DEVICES=0 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --leaves-log2 20 > gpurun_out/bench_r02_dup.json 2> gpurun_out/bench_r02_dup.err; echo "two ranks on one physical GPU rc=$? (must be non-zero)"; wc -c gpurun_out/bench_r02_dup.json; grep -i -h "duplicate\|invalid usage\|ncclInvalid\|Error" gpurun_out/bench_r02_dup.err | head -5
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_r02_c.log 2>&1; tail -3 gpurun_out/pytest_r02_c.log
