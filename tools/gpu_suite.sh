cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_r02_final.log 2>&1; tail -14 gpurun_out/pytest_r02_final.log
python -c "import __graft_entry__ as g; g.smoke()"
