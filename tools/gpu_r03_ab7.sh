cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
E=$GRAFT_REPO_ROOT/build/ab
bash tools/ab_env.sh default: exp14_3wg:VKMR_HIP_LIB=$E/libexp.so,VKMR_MAP_VARIANT=14 exp15_4wg:VKMR_HIP_LIB=$E/libexp.so,VKMR_MAP_VARIANT=15 exp7:VKMR_HIP_LIB=$E/libexp.so,VKMR_MAP_VARIANT=7 > gpurun_out/r03/ab7.txt 2>&1; cat gpurun_out/r03/ab7.txt
