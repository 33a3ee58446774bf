cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
E=$GRAFT_REPO_ROOT/build/ab
for round in 1 2 3; do for lib in default $E/libsplit0.so $E/libsplit3.so $E/libsplit5.so $E/libsplit6.so $E/libsplit8.so; do
  if [ "$lib" = default ]; then unset VKMR_HIP_LIB; else export VKMR_HIP_LIB=$lib; fi
  echo "$(basename $lib) $(python3 tools/reduce_probe.py 26 30 | cut -d';' -f1) | $(python3 tools/long_strings_probe.py 24 127 | sed 's/.*ms per launch/map 2^24 ms/' | cut -d' ' -f1-4)"
done; done > gpurun_out/r03/split_sweep.txt 2>&1
cat gpurun_out/r03/split_sweep.txt
