cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
E=$GRAFT_REPO_ROOT/build/ab
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reduce or proof or slices" > gpurun_out/r03/pytest_pf.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_pf.log
bash tools/ab_env.sh default: prev:VKMR_HIP_LIB=$E/libprev.so > gpurun_out/r03/ab4.txt 2>&1; cat gpurun_out/r03/ab4.txt
