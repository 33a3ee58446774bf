cd $GRAFT_REPO_ROOT
python3 tools/map_stamps.py 23 127 > gpurun_out/map_stamps_r02.txt 2>&1; cat gpurun_out/map_stamps_r02.txt
for bl in 23 24 25 26; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --batch-log2 $bl --no-cpu-baseline --no-pipeline --no-long-strings 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); v = d['valu_roofline']
print('batch-log2 $bl', 'ms/step', round(d['ms_per_step'], 3), 'map', round(v['map_ms_per_step'], 3), 'reduce', round(v['reduce_ms_per_step'], 3), 'map T/s', round(v['map_achieved_tops'],2), d['root_matches_golden'])"
done
# steady state of the stream processor: 256 slices of 2^12 through vkmr hip:0 -- how many hipMalloc/hipFree?
vk_merkle_roots_amd/bin/rndm 42 1048576 127 > /tmp/g3.txt
( cd /tmp && export TMPDIR=/tmp && VKMR_SLICE_LOG2=12 VKMR_VERBOSE=0 timeout -k 10 200 rocprofv3 --hip-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hiptrace_r02 -- $GRAFT_REPO_ROOT/vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g3.txt > $GRAFT_REPO_ROOT/gpurun_out/hiptrace_r02.out 2> $GRAFT_REPO_ROOT/gpurun_out/hiptrace_r02.err )
tail -2 gpurun_out/hiptrace_r02.out
cat gpurun_out/hiptrace_r02/*/*hip_api_stats.csv 2>/dev/null | head -30
