# round 3 measurement set (GPU box): PMC passes over bench.py and the long-string probe -> profiles/pmc_latest.json (keyed on the
# build id of the library profiled); rocprofv3 kernel stats of the same command; the bench line; vkmr from a file / a pipe with the
# process wall clock; a hip-trace of vkmr hip:0 (does copy k+1 overlap map k?)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash tools/pmc_profile.sh r03 > gpurun_out/r03/pmc.log 2>&1
( cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r03_long/$c -- python3 $GRAFT_REPO_ROOT/tools/long_strings_probe.py > /dev/null 2>&1; done )
python3 tools/pmc_to_json.py gpurun_out/pmc_r03 gpurun_out/r03/pmc_r03.json --long-strings-dir gpurun_out/pmc_r03_long > /dev/null
cp gpurun_out/r03/pmc_r03.json profiles/pmc_latest.json
cp gpurun_out/pmc_r03/summary.txt gpurun_out/r03/pmc_summary.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-clock-leg > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03/prof_bench.err )
find gpurun_out/prof_r03 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03/kernel_stats.csv \;
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r03/bench.json').read().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','root_matches_golden']}); print(d['roofline']); print(d['roofline_reduce']); print(d['long_strings']['roofline'])"
# end to end from a file and a pipe
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
echo "# vkmr hip:0 < file (2^25 strings, 2.13 GB), the program's own line"
for i in 1 2 3 4; do vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>/dev/null | tail -1; done
echo "# cat file | vkmr hip:0"
for i in 1 2; do cat /tmp/g25.txt | vk_merkle_roots_amd/bin/vkmr hip:0 2>/dev/null | tail -1; done
echo "# process wall clock (HIP start-up and teardown included), vkmr hip:0 < file"
for i in 1 2 3 4; do python3 -c "
import subprocess, time
t=time.time(); r=subprocess.run(['vk_merkle_roots_amd/bin/vkmr','hip:0'], stdin=open('/tmp/g25.txt','rb'), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL); w=time.time()-t
line=[l for l in r.stdout.decode().splitlines() if 'computed root' in l][-1]
print('process wall %.3f s; printed %s ms' % (w, line.rsplit(' in ',1)[1]))"; done
echo "# the same with VKMR_ORDERLY_EXIT=1 (destructors run)"
for i in 1 2; do python3 -c "
import subprocess, time, os
t=time.time(); r=subprocess.run(['vk_merkle_roots_amd/bin/vkmr','hip:0'], stdin=open('/tmp/g25.txt','rb'), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, VKMR_ORDERLY_EXIT='1')); w=time.time()-t
line=[l for l in r.stdout.decode().splitlines() if 'computed root' in l][-1]
print('process wall %.3f s; printed %s ms' % (w, line.rsplit(' in ',1)[1]))"; done
echo "# vkmr CPU on 2^20 strings: wall clock, and HIP calls made (rocprofv3 --hip-trace)"
vk_merkle_roots_amd/bin/rndm 42 1048576 127 > /tmp/g20.txt 2>/dev/null
python3 -c "
import subprocess, time
t=time.time(); r=subprocess.run(['vk_merkle_roots_amd/bin/vkmr','CPU'], stdin=open('/tmp/g20.txt','rb'), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL); w=time.time()-t
print('process wall %.3f s;' % w, r.stdout.decode().splitlines()[-1])"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/cpu_trace -- $GRAFT_REPO_ROOT/vk_merkle_roots_amd/bin/vkmr CPU < /tmp/g20.txt > /dev/null 2>&1; echo "hip api calls traced for vkmr CPU: $(find /tmp/cpu_trace -name '*hip_api_trace.csv' -exec cat {} \; | grep -vc Domain)" )
} > gpurun_out/r03/end_to_end.txt 2>&1
cat gpurun_out/r03/end_to_end.txt
# copy/kernel overlap in the front end
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_r03 -- $GRAFT_REPO_ROOT/vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1 )
python3 tools/overlap_from_trace.py gpurun_out/trace_r03 > gpurun_out/r03/copy_map_overlap.txt 2>&1; cat gpurun_out/r03/copy_map_overlap.txt
