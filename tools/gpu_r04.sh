#!/bin/bash
# tools/gpu_r04.sh -- round 4's gpurun command lists, one parameterised runner (GPU box): bash tools/gpu_r04.sh <task> [args...]
# Output goes to gpurun_out/r04/; what is to be judged is copied to profiles/r04_* afterwards.
#   variants <v...>        parity of the listed VKMR_MAP_VARIANT values (tests/test_gpu_random.py), then interleaved bench A/B of them
#   ab <label:ENV=..>...   interleaved bench A/B (tools/ab_env.sh)
#   clock <log2> [variant] tools/kernel_clock.py on the stamped library (product twin, or the experiments twin with a variant)
#   suite                  the GPU test suite + smoke
#   bench [args]           one bench line
#   measure                PMC passes -> profiles/pmc_latest.json, rocprofv3 kernel stats of the bench command, the bench line
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r04
mkdir -p $OUT
task=$1; shift
case $task in
variants)
  sel=$(printf "%s or " "$@"); sel=${sel% or }
  timeout -k 10 900 python -m pytest tests/test_gpu_random.py -m gpu -q -k "fetch_mode and ($sel)" > $OUT/pytest_variants.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_variants.log
  specs="default:"; for v in "$@"; do [ "$v" != 0 ] && specs="$specs exp$v:VKMR_HIP_LIB=$PWD/build/ab/libexp.so,VKMR_MAP_VARIANT=$v"; done
  bash tools/ab_env.sh $specs | tee -a $OUT/ab_variants.txt
  ;;
ab)
  bash tools/ab_env.sh "$@" | tee -a $OUT/ab.txt
  ;;
clock)
  log2=$1; variant=$2
  if [ -n "$variant" ]; then
    VKMR_MAP_VARIANT=$variant timeout -k 10 300 python3 tools/kernel_clock.py --leaves-log2 $log2 --lib build/ab/libexp_stamps.so > $OUT/kernel_clock_${log2}_v$variant.json 2> $OUT/kernel_clock_${log2}_v$variant.err; echo "kernel_clock v$variant rc=$?"
    tail -c 1200 $OUT/kernel_clock_${log2}_v$variant.json
  else
    timeout -k 10 300 python3 tools/kernel_clock.py --leaves-log2 $log2 > $OUT/kernel_clock_$log2.json 2> $OUT/kernel_clock_$log2.err; echo "kernel_clock rc=$?"
    tail -c 1500 $OUT/kernel_clock_$log2.json; tail -12 $OUT/kernel_clock_$log2.err
  fi
  ;;
suite)
  timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc=$?"
  ;;
bench)
  timeout -k 10 600 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench.json
  ;;
measure)
  bash tools/pmc_profile.sh r04 > $OUT/pmc.log 2>&1
  ( cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r04_long/$c -- python3 $GRAFT_REPO_ROOT/tools/long_strings_probe.py > /dev/null 2>&1; done )
  python3 tools/pmc_to_json.py gpurun_out/pmc_r04 $OUT/pmc_r04.json --long-strings-dir gpurun_out/pmc_r04_long > /dev/null
  cp $OUT/pmc_r04.json profiles/pmc_latest.json
  cp gpurun_out/pmc_r04/summary.txt $OUT/pmc_summary.txt
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-clock-leg > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof_bench.err )
  find gpurun_out/prof_r04 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
  timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  ;;
*) echo "unknown task $task"; exit 2;;
esac
