cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
echo "# VKMR_TIMING=1 vkmr hip:0 < file (2^25 strings)"
for i in 1 2 3; do python3 -c "
import subprocess, time, os
t=time.time(); r=subprocess.run(['vk_merkle_roots_amd/bin/vkmr','hip:0'], stdin=open('/tmp/g25.txt','rb'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, VKMR_TIMING='1')); w=time.time()-t
line=[l for l in r.stdout.decode().splitlines() if 'computed root' in l][-1]
print('process wall %.3f s; printed %s ms;' % (w, line.rsplit(' in ',1)[1]), ' | '.join(l for l in r.stderr.decode().splitlines() if 'timing' in l))"; done
echo "# vkmr hip:0 < file of 2^26 strings (4.3 GB): printed, process wall"
vk_merkle_roots_amd/bin/rndm 42 67108864 127 > /tmp/g26.txt 2>/dev/null
for i in 1 2 3 4; do python3 -c "
import subprocess, time
t=time.time(); r=subprocess.run(['vk_merkle_roots_amd/bin/vkmr','hip:0'], stdin=open('/tmp/g26.txt','rb'), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL); w=time.time()-t
line=[l for l in r.stdout.decode().splitlines() if 'computed root' in l][-1]
print('process wall %.3f s; %s' % (w, line))"; done
rm -f /tmp/g26.txt
echo "# cat file | vkmr hip:0"
for i in 1 2 3; do cat /tmp/g25.txt | vk_merkle_roots_amd/bin/vkmr hip:0 2>/dev/null | tail -1; done
echo "# cat file > /dev/null (the pipe alone)"
python3 -c "
import subprocess, time
t=time.time(); p=subprocess.Popen(['cat','/tmp/g25.txt'], stdout=subprocess.PIPE); n=0
while True:
    b=p.stdout.read(1<<24)
    if not b: break
    n+=len(b)
print('python reads %d bytes from the pipe in %.3f s' % (n, time.time()-t))"
echo "# hip api calls of vkmr CPU (rocprofv3 --hip-trace)"
vk_merkle_roots_amd/bin/rndm 42 1048576 127 > /tmp/g20.txt 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --hip-trace --output-format csv -d /tmp/cpu_trace -- $GRAFT_REPO_ROOT/vk_merkle_roots_amd/bin/vkmr CPU < /tmp/g20.txt > /dev/null 2>&1; find /tmp/cpu_trace -name '*hip_api_trace.csv' -exec cut -d, -f1-3 {} \; | sort | uniq -c | sort -rn | head )
} > gpurun_out/r03/end_to_end2.txt 2>&1
cat gpurun_out/r03/end_to_end2.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/fe_trace -- $GRAFT_REPO_ROOT/vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1; find /tmp/fe_trace -name '*hip_api_stats.csv' -exec head -16 {} \; ) > gpurun_out/r03/frontend_api_stats.txt 2>&1; cat gpurun_out/r03/frontend_api_stats.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_r03 -- $GRAFT_REPO_ROOT/vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1 )
python3 tools/overlap_from_trace.py gpurun_out/trace_r03 > gpurun_out/r03/copy_map_overlap.txt 2>&1; cat gpurun_out/r03/copy_map_overlap.txt
