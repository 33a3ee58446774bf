cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
E=$GRAFT_REPO_ROOT/build/ab
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > gpurun_out/r03/pytest_pad.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_pad.log
bash tools/ab_env.sh default: split2:VKMR_HIP_LIB=$E/libsplit2.so split3:VKMR_HIP_LIB=$E/libsplit3.so split4:VKMR_HIP_LIB=$E/libsplit4.so > gpurun_out/r03/ab3.txt 2>&1; cat gpurun_out/r03/ab3.txt
