cd $GRAFT_REPO_ROOT
VKMR_MAP_VARIANT=5 timeout -k 10 600 python -m pytest tests/test_gpu_random.py tests/test_gpu_parity.py -q -x -k "map or fetch_mode" > gpurun_out/pytest_mode4b.log 2>&1; tail -2 gpurun_out/pytest_mode4b.log
for ml in 400 1200 4096; do for v in 0 4 0; do VKMR_MAP_VARIANT=$v python3 tools/long_strings_probe.py 21 $ml; done; done
