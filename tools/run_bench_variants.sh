set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r02_a.json 2> gpurun_out/bench_r02_a.err
echo "rc=$?"
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --slice-log2 23 --no-cpu-baseline --no-pipeline --no-long-strings > gpurun_out/bench_r02_s23.json 2> gpurun_out/bench_r02_s23.err
echo "rc=$?"
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline --no-pipeline --no-long-strings > gpurun_out/bench_r02_dist1.json 2> gpurun_out/bench_r02_dist1.err
echo "rc=$?"
timeout -k 10 100 python bench.py --gpus 2 --steps 2 > gpurun_out/bench_r02_g2.json 2> gpurun_out/bench_r02_g2.err
echo "rc(gpus 2 on one GPU, must be non-zero)=$?"
timeout -k 10 300 python bench.py --gpus 2 --rehearse-gloo --steps 3 --warmup 1 > gpurun_out/bench_r02_gloo2.json 2> gpurun_out/bench_r02_gloo2.err
echo "rc=$?"
