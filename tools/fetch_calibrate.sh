#!/bin/bash
# tools/fetch_calibrate.sh -- FETCH_SIZE / TCC_EA0_RDREQ calibration on known byte counts (GPU box only).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/fetch_calibrate
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $REPO/tools/fetch_calibrate > $OUT/fetch.log 2>&1 || echo "fetch pass failed" >> $OUT/fail.log
timeout -k 10 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/rdreq -- $REPO/tools/fetch_calibrate > $OUT/rdreq.log 2>&1 || echo "rdreq pass failed" >> $OUT/fail.log
timeout -k 10 120 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- $REPO/tools/fetch_calibrate > $OUT/tcc.log 2>&1 || echo "tcc pass failed" >> $OUT/fail.log
python3 $REPO/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
grep -i "TCC_EA0_RD\|TCC_EA0_WR" $OUT/counters.txt | head -40
