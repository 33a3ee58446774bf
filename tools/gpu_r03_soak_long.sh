cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 380 python3 tests/soak/soak_abi.py 300 > gpurun_out/r03/soak_abi_long.txt 2>&1; tail -2 gpurun_out/r03/soak_abi_long.txt
timeout -k 10 400 python3 tests/soak/soak_frontend.py 300 > gpurun_out/r03/soak_frontend_long.txt 2>&1; tail -2 gpurun_out/r03/soak_frontend_long.txt
