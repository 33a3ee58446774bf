cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/walk_calibrate
mkdir -p $OUT
tools/fetch_calibrate 4096 > $OUT/timing.txt 2>&1; cat $OUT/timing.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $GRAFT_REPO_ROOT/tools/fetch_calibrate 4096 > $OUT/fetch.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A3 "calib_walk" $OUT/summary.txt
