cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tests/soak/soak_abi.py 200 > gpurun_out/soak_r02d.txt 2>&1; tail -1 gpurun_out/soak_r02d.txt
timeout -k 10 400 python3 tests/soak/soak_frontend.py 240 > gpurun_out/soak_frontend_r02d.txt 2>&1; tail -1 gpurun_out/soak_frontend_r02d.txt
