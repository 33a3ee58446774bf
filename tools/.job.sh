cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
E=$PWD/build/ab/libexp.so
bash tools/gpu_r04.sh ab default: st0:VKMR_HIP_LIB=$E,VKMR_MAP_VARIANT=23,VKMR_MAP_STAGGER=0 st3:VKMR_HIP_LIB=$E,VKMR_MAP_VARIANT=23,VKMR_MAP_STAGGER=3 st5:VKMR_HIP_LIB=$E,VKMR_MAP_VARIANT=23,VKMR_MAP_STAGGER=5 st8:VKMR_HIP_LIB=$E,VKMR_MAP_VARIANT=23,VKMR_MAP_STAGGER=8
VKMR_MAP_STAGGER=5 bash tools/gpu_r04.sh clock 24 23
VKMR_MAP_STAGGER=5 bash tools/gpu_r04.sh clock 26 23
bash tools/gpu_r04.sh clock 26 7
bash tools/gpu_r04.sh clock 26
VKMR_HIP_LIB=$E VKMR_MAP_VARIANT=23 timeout -k 10 200 python3 tools/diag_variant.py 5 3000 4096 6 9000 700 8 150000 300 > gpurun_out/r04/diag23.txt 2>&1; cat gpurun_out/r04/diag23.txt
