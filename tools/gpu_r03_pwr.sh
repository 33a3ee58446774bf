cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 200 python3 tools/kernel_clock.py --leaves-log2 24 > gpurun_out/r03/kernel_clock_pwr.json 2> gpurun_out/r03/kernel_clock_pwr.err; echo "kernel_clock rc=$?"; cat gpurun_out/r03/kernel_clock_pwr.json
timeout -k 10 200 ./tools/clock_probe 1.0 > gpurun_out/r03/clock_probe3.txt 2>&1; echo "clock_probe rc=$?"; cut -c1-40,150-400 gpurun_out/r03/clock_probe3.txt
