#!/bin/bash
# tools/pmc_profile.sh -- rocprofv3 counter passes over a short bench run (GPU box only).
# Each --pmc set runs in its own process (SQ: 8 slots, TCC: 4; FETCH_SIZE takes 3, WRITE_SIZE 2).
# Usage (from the repo root on the box):  bash tools/pmc_profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-run}; shift
ARGS=${@:---leaves-log2 26 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-long-strings --no-clock-leg}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/bench.py $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/fail.log
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum     # request-size split: every read request of gfx950 is 128 B (32B = 0)
python3 $REPO/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
