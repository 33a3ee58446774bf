# round 3, first GPU call: in-kernel clock probes, parity of the changed kernels, one bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 200 ./tools/clock_probe 2.0 > gpurun_out/r03/clock_probe.txt 2>&1; echo "clock_probe rc=$?"
timeout -k 10 200 python3 tools/kernel_clock.py --leaves-log2 24 > gpurun_out/r03/kernel_clock_24.json 2> gpurun_out/r03/kernel_clock_24.err; echo "kernel_clock rc=$?"
timeout -k 10 300 python3 tools/kernel_clock.py --leaves-log2 26 > gpurun_out/r03/kernel_clock_26.json 2> gpurun_out/r03/kernel_clock_26.err; echo "kernel_clock26 rc=$?"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > gpurun_out/r03/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03/pytest_a.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_a.json 2> gpurun_out/r03/bench_a.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r03/kernel_clock_26.json
