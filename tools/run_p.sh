cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tests/soak/soak_abi.py 240 > gpurun_out/soak_r02b.txt 2>&1; tail -2 gpurun_out/soak_r02b.txt
timeout -k 10 500 python3 tests/soak/soak_frontend.py 360 > gpurun_out/soak_frontend_r02b.txt 2>&1; tail -3 gpurun_out/soak_frontend_r02b.txt
