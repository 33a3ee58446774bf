# north star's "message schedule and round constants staged in LDS", timed INSIDE map_kernel on config 3 (experiments build):
# variant 11 = shipped compression (K literals, ring in VGPRs) with the same 17600-word staging; 9 = K[64] read from LDS;
# 10 = K and the 16-word schedule ring in LDS (113 KiB of LDS: one workgroup per CU); default = the product library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
E=$GRAFT_REPO_ROOT/build/ab
timeout -k 10 600 python -m pytest tests/test_gpu_random.py -m gpu -x -q -k "fetch_mode" > gpurun_out/r03/pytest_variants.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_variants.log
bash tools/ab_env.sh default: v11_registers:VKMR_HIP_LIB=$E/libexp.so,VKMR_MAP_VARIANT=11 v9_K_in_LDS:VKMR_HIP_LIB=$E/libexp.so,VKMR_MAP_VARIANT=9 v10_K_and_ring_in_LDS:VKMR_HIP_LIB=$E/libexp.so,VKMR_MAP_VARIANT=10 > gpurun_out/r03/map_lds_schedule_ab.txt 2>&1; cat gpurun_out/r03/map_lds_schedule_ab.txt
