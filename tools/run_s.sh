cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_frontend.py -m gpu -q -k "config3_from_stdin" > gpurun_out/pytest_cfg3_stdin.log 2>&1; tail -3 gpurun_out/pytest_cfg3_stdin.log
