cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
E=$GRAFT_REPO_ROOT/build/ab
bash tools/ab_env.sh default: twolevel:VKMR_HIP_LIB=$E/libtwolevel.so default: twolevel:VKMR_HIP_LIB=$E/libtwolevel.so > gpurun_out/r03/ab6.txt 2>&1; cat gpurun_out/r03/ab6.txt
