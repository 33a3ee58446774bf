cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_frontend.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1
for round in 1 2 3 4 5 6; do
for v in "VKMR_DEVICE_SPLIT=0" "VKMR_DEVICE_SPLIT=1"; do
  env $v VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | python3 -c "
import sys,re
t={}
for l in sys.stdin:
    m=re.match(r'\[timing\] (.+): ([0-9.e+-]+) ms',l)
    if m: t[m.group(1).strip()]=float(m.group(2))
    m=re.search(r'=> ([0-9a-f]{8}).* in ([0-9.]+)\$',l)
    if m and 'computed root' in l: t['printed']=float(m.group(2)); t['root']=m.group(1)
print('%-22s root %s printed %6.1f  pass 1 %5.1f  pass 2 / copy+count %5.1f  pipeline-full wait %5.1f' % ('$v', t.get('root'), t['printed'], t['pack pass 1 (index the lines, fork-join)'], t['pack pass 2 (copy the lines, fork-join)'], t['wait for the oldest mapping (pipeline full)']))"
done; done
} > gpurun_out/r03/split_ab.txt 2>&1
cat gpurun_out/r03/split_ab.txt
