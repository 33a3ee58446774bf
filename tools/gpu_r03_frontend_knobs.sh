cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
summ() { python3 -c "
import sys,re
t={}
for l in sys.stdin:
    m=re.match(r'\[timing\] (.+): ([0-9.e+-]+) ms',l)
    if m: t[m.group(1).strip()]=float(m.group(2))
    m=re.search(r' in ([0-9.]+)\$',l)
    if m and 'computed root' in l: t['printed']=float(m.group(1))
g=lambda k: t.get(k,float('nan'))
print('%-40s enumerate %6.1f construct %6.1f printed %6.1f total %6.1f | index %5.1f pack %5.1f serial %4.1f | batch wait %5.1f map wait %5.1f dispatch %5.1f (copies %4.1f launch %4.1f) drain %4.1f+%4.1f' % ('$1', g('devices enumerated'), g('backend constructed')-g('devices enumerated'), g('printed'), g('root printed'), g('pack pass 1 (index the lines, fork-join)'), g('pack pass 2 (copy the lines, fork-join)'), g('pack, serial remainder'), g('wait for / allocate a batch'), g('wait for the oldest mapping (pipeline full)'), g('map dispatch (copies + launch)'), g('of which the two copies'), g('of which the launch'), g('drain: last batch and mappings'), g('drain: reductions and root')))"; }
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1
for round in 1 2 3 4; do
for v in "VKMR_TIMING=1" "VKMR_BATCH_MB=64" "VKMR_BATCH_MB=64 VKMR_MAX_INFLIGHT=4" "VKMR_MAX_INFLIGHT=2" "VKMR_MAX_INFLIGHT=4" "VKMR_BATCH_MB=36"; do
  env $v VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | summ "$v"
done; done
} > gpurun_out/r03/frontend4.txt 2>&1
cat gpurun_out/r03/frontend4.txt
