cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for r in 1 2; do
BENCH_ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-long-strings --no-clock-leg" bash -c 'python bench.py $BENCH_ARGS 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); v=d[\"valu_roofline\"]; print(\"wave  \", round(d[\"ms_per_step\"],3), round(v[\"map_ms_per_step\"],3), round(v[\"reduce_ms_per_step\"],3), d[\"root_matches_golden\"])"'
BENCH_ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-long-strings --no-clock-leg --levels-variant" bash -c 'python bench.py $BENCH_ARGS 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); v=d[\"valu_roofline\"]; print(\"levels\", round(d[\"ms_per_step\"],3), round(v[\"map_ms_per_step\"],3), round(v[\"reduce_ms_per_step\"],3), d[\"root_matches_golden\"])"'
done > gpurun_out/r03/levels_variant.txt 2>&1
cat gpurun_out/r03/levels_variant.txt
