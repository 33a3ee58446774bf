cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1
for round in 1 2 3; do
for v in "VKMR_SETUP=0" "VKMR_SETUP=1" "VKMR_SETUP=2" "VKMR_SETUP=3" "VKMR_SETUP=4" "VKMR_SETUP=1 VKMR_SHARE_REDUCE_STREAM=1" "VKMR_SETUP=2 VKMR_SHARE_REDUCE_STREAM=1" "VKMR_SETUP=4 VKMR_SHARE_REDUCE_STREAM=1"; do
  env $v VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | python3 -c "
import sys,re
t={}
for l in sys.stdin:
    m=re.match(r'\[timing\] (.+): ([0-9.e+-]+) ms',l)
    if m: t[m.group(1)]=float(m.group(2))
    m=re.search(r' in ([0-9.]+)\$',l)
    if m and 'computed root' in l: t['printed']=float(m.group(1))
print('%-48s enumerate %6.1f construct %6.1f printed %6.1f total %6.1f | batch wait %5.1f map wait %5.1f dispatch %5.1f drain %4.1f+%4.1f' % ('$v', t['devices enumerated'], t['backend constructed']-t['devices enumerated'], t['printed'], t['root printed'], t['wait for / allocate a batch'], t['wait for the oldest mapping (pipeline full)'], t['map dispatch (copies + launch)'], t['drain: last batch and mappings'], t['drain: reductions and root']))"
done; done
} > gpurun_out/r03/frontend3.txt 2>&1
cat gpurun_out/r03/frontend3.txt
