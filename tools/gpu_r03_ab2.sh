cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
E=$GRAFT_REPO_ROOT/build/ab
bash tools/ab_env.sh default: m3:VKMR_HIP_LIB=$E/libm3.so m2:VKMR_HIP_LIB=$E/libm2.so l3:VKMR_HIP_LIB=$E/libl3.so exp7:VKMR_HIP_LIB=$E/libexp.so,VKMR_MAP_VARIANT=7 exp12:VKMR_HIP_LIB=$E/libexp.so,VKMR_MAP_VARIANT=12 exp13:VKMR_HIP_LIB=$E/libexp.so,VKMR_MAP_VARIANT=13 exp4:VKMR_HIP_LIB=$E/libexp.so,VKMR_MAP_VARIANT=4 exp1:VKMR_HIP_LIB=$E/libexp.so,VKMR_MAP_VARIANT=1 > gpurun_out/r03/ab2.txt 2>&1; cat gpurun_out/r03/ab2.txt
