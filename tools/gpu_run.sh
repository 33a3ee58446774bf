#!/bin/bash
# tools/gpu_run.sh -- THE gpurun command list, one parameterised runner (GPU box): [ROUND=r04] bash tools/gpu_run.sh <task> [args...]
# (Rounds 2 and 3 kept one throw-away script per call -- 22 of them; their tasks are the cases below.)
# Output goes to gpurun_out/$ROUND/; what is to be judged is copied to profiles/${ROUND}_* afterwards.
#   variants <v...>        parity of the listed VKMR_MAP_VARIANT values (tests/test_gpu_random.py), then interleaved bench A/B of them
#   ab <label:ENV=..>...   interleaved bench A/B (tools/ab_env.sh)
#   clock <log2> [variant] tools/kernel_clock.py on the stamped library (product twin, or the experiments twin with a variant)
#   suite                  the GPU test suite + smoke
#   bench [args]           one bench line
#   measure                PMC passes -> profiles/pmc_latest.json, rocprofv3 kernel stats of the bench command, the bench line
#   e2e [log2...]          `vkmr hip:0 < file` of 2^k strings (default 25 26): printed time and process wall, N runs each, the distribution
#   frontend               VKMR_TIMING=1 phases, a pipe, hip-api stats and copy/kernel overlap of `vkmr hip:0`
#   rehearsals             soaks against the oracle; torchrun --nproc 2 (gloo rehearsal; plain N=2 must fail on one GPU); --force-dist (RCCL, one rank)
#   cumask                 tools/cu_mask_probe.py: where CU-mask bits land, map/reduce on half the CUs with and without neighbours
#   issue <set>            tools/issue_patterns (python3 tools/gen_issue_patterns.py <set> and a build beforehand)
#   proofs                 tools/proof_timing.py: a slice reduced with and without proofs written in the pass
cd ${GRAFT_REPO_ROOT:-.}
ROUND=${ROUND:-r04}
OUT=gpurun_out/$ROUND
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
  bash tools/pmc_profile.sh $ROUND > $OUT/pmc.log 2>&1
  ( cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${ROUND}_long/$c -- python3 $GRAFT_REPO_ROOT/tools/long_strings_probe.py > /dev/null 2>&1; done )
  python3 tools/pmc_to_json.py gpurun_out/pmc_$ROUND $OUT/pmc_$ROUND.json --long-strings-dir gpurun_out/pmc_${ROUND}_long > /dev/null
  cp $OUT/pmc_$ROUND.json profiles/pmc_latest.json
  cp gpurun_out/pmc_$ROUND/summary.txt $OUT/pmc_summary.txt
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$ROUND -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-clock-leg > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof_bench.err )
  find gpurun_out/prof_$ROUND -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
  timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  ;;
e2e)
  [ $# -eq 0 ] && set -- 25 26
  for k in "$@"; do
    vk_merkle_roots_amd/bin/rndm 42 $((1 << k)) 127 > /tmp/g$k.txt 2>/dev/null
    vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g$k.txt > /dev/null 2>&1
    python3 - $k ${E2E_RUNS:-12} <<'PY'
import subprocess, sys, time, statistics
k, n = int(sys.argv[1]), int(sys.argv[2])
printed, wall = [], []
for _ in range(n):
    t = time.time()
    r = subprocess.run(["vk_merkle_roots_amd/bin/vkmr", "hip:0"], stdin=open(f"/tmp/g{k}.txt", "rb"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    wall.append(time.time() - t)
    line = [l for l in r.stdout.decode().splitlines() if "computed root" in l][-1]
    printed.append(float(line.rsplit(" in ", 1)[1]))
q = statistics.quantiles(printed, n=4)
print(f"vkmr hip:0 < 2^{k} strings, {n} runs: printed min {min(printed):.1f}  quartiles {q[0]:.1f} / {q[1]:.1f} / {q[2]:.1f}  max {max(printed):.1f} ms; process wall median {statistics.median(wall):.3f} s (min {min(wall):.3f})")
print("  printed:", " ".join(f"{p:.1f}" for p in printed))
PY
    rm -f /tmp/g$k.txt
  done 2>&1 | tee $OUT/end_to_end.txt
  ;;
frontend)
  {
  vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
  echo "# VKMR_TIMING=1 vkmr hip:0 < file (2^25 strings)"
  for i in 1 2 3; do python3 -c "
import subprocess, time, os
t=time.time(); r=subprocess.run(['vk_merkle_roots_amd/bin/vkmr','hip:0'], stdin=open('/tmp/g25.txt','rb'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, VKMR_TIMING='1')); w=time.time()-t
line=[l for l in r.stdout.decode().splitlines() if 'computed root' in l][-1]
print('process wall %.3f s; printed %s ms;' % (w, line.rsplit(' in ',1)[1]), ' | '.join(l for l in r.stderr.decode().splitlines() if 'timing' in l))"; done
  echo "# cat file | vkmr hip:0"
  for i in 1 2 3; do cat /tmp/g25.txt | vk_merkle_roots_amd/bin/vkmr hip:0 2>/dev/null | tail -1; done
  } > $OUT/frontend.txt 2>&1; cat $OUT/frontend.txt
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/fe_trace -- $GRAFT_REPO_ROOT/vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1; find /tmp/fe_trace -name '*hip_api_stats.csv' -exec head -16 {} \; ) > $OUT/frontend_api_stats.txt 2>&1
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$ROUND -- $GRAFT_REPO_ROOT/vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1 )
  python3 tools/overlap_from_trace.py gpurun_out/trace_$ROUND > $OUT/copy_map_overlap.txt 2>&1; cat $OUT/copy_map_overlap.txt
  ;;
rehearsals)
  timeout -k 10 300 python3 tests/soak/soak_abi.py ${SOAK_ABI:-150} > $OUT/soak_abi.txt 2>&1; tail -2 $OUT/soak_abi.txt
  timeout -k 10 400 python3 tests/soak/soak_frontend.py ${SOAK_FRONTEND:-200} > $OUT/soak_frontend.txt 2>&1; tail -3 $OUT/soak_frontend.txt
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse-gloo --steps 3 --warmup 1 --no-config5 --no-long-strings --cpu-sample-log2 20 > $OUT/bench_torchrun_gloo2.json 2> $OUT/bench_torchrun_gloo2.err; echo "torchrun rehearsal rc=$?"; python3 -c "
import json
d=json.loads(open('$OUT/bench_torchrun_gloo2.json').read().splitlines()[-1]); print({k:d.get(k) for k in ('n_gpus','ms_per_step','root_matches_golden','sub_roots_match_golden')}, d['config']['collective'], 'cpu_baseline' in d, 'valu_roofline' in d)"
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 2 > $OUT/bench_torchrun_2.json 2> $OUT/bench_torchrun_2.err; echo "torchrun N=2 on one GPU rc=$? (must be non-zero)"; wc -c $OUT/bench_torchrun_2.json
  timeout -k 10 200 python bench.py --force-dist --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline --no-long-strings --no-clock-leg > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err; echo "force-dist rc=$?"; python3 -c "
import json
d=json.loads(open('$OUT/bench_force_dist.json').read().splitlines()[-1]); print(d['config']['rccl'], d['gather_ms'], d['root_matches_golden'])"
  ;;
issue)
  timeout -k 10 600 ./tools/issue_patterns ${2:-2.0} > $OUT/issue_patterns_$1.txt 2>&1; tail -40 $OUT/issue_patterns_$1.txt
  ;;
proofs)
  for k in "26 8" "26 1" "26 16" "23 16"; do set -- $k; timeout -k 10 300 python3 tools/proof_timing.py --log2 $1 --proofs $2; done > $OUT/proof_timing.txt 2>&1; cat $OUT/proof_timing.txt
  ;;
cumask)
  hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/libwhere.so tools/where.hip
  timeout -k 10 400 python3 tools/cu_mask_probe.py > $OUT/cu_mask_probe.txt 2>&1; tail -14 $OUT/cu_mask_probe.txt
  ;;
*) echo "unknown task $task"; exit 2;;
esac
