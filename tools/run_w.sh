cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 tests/soak/soak_frontend.py 420 > gpurun_out/soak_frontend_r02c.txt 2>&1; tail -3 gpurun_out/soak_frontend_r02c.txt
