cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export VKMR_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/libexp.so
for round in 1 2; do
for spec in "21 4096" "22 1024" "23 400"; do
  for v in 0 4 5 3; do
    VKMR_MAP_VARIANT=$v python3 tools/long_strings_probe.py $spec | sed "s/^/variant $v: /"
  done
done; done > gpurun_out/r03/long_strings_modes.txt 2>&1
cat gpurun_out/r03/long_strings_modes.txt
