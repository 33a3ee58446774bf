cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash tools/ab_libs.sh build/ab/libplain.so build/ab/libprio1.so > gpurun_out/r03/ab_prio2.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/r03/ab_prio2.txt
timeout -k 10 200 python3 tools/kernel_clock.py --leaves-log2 24 > gpurun_out/r03/kernel_clock_24_prio.json 2> gpurun_out/r03/kernel_clock_24_prio.err; echo "kernel_clock rc=$?"; cat gpurun_out/r03/kernel_clock_24_prio.json
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > gpurun_out/r03/pytest_prio.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_prio.log
