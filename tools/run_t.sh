cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_frontend.py -m gpu -q > gpurun_out/pytest_frontend2.log 2>&1; tail -5 gpurun_out/pytest_frontend2.log
