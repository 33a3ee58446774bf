cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
E=$GRAFT_REPO_ROOT/build/ab
for v in 7 12 4 13; do
  VKMR_MAP_VARIANT=$v timeout -k 10 200 python3 tools/kernel_clock.py --leaves-log2 24 --lib $E/libexp_stamps.so 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); m=d['map_kernel']; r=d['reduce_pass_kernel']
print('variant $v map', m['GHz_median'], 'GHz', m['ms_per_launch_sustained'], 'ms', m['phase_share'], '| reduce', r['GHz_median'], 'GHz', r['ms_per_launch_sustained'], 'ms')"
done > gpurun_out/r03/clk_variants.txt 2>&1
timeout -k 10 200 python3 tools/kernel_clock.py --leaves-log2 24 --lib $E/libm2_stamps.so 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readlines()[-1]); m=d['map_kernel']; r=d['reduce_pass_kernel']
print('m2 map', m['GHz_median'], 'GHz', m['ms_per_launch_sustained'], 'ms | reduce', r['GHz_median'], 'GHz', r['ms_per_launch_sustained'], 'ms')" >> gpurun_out/r03/clk_variants.txt 2>&1
cat gpurun_out/r03/clk_variants.txt
