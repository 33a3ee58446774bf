cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1
for round in 1 2 3 4 5 6 7 8 9 10; do
for v in "VKMR_PACK_STREAM=0" "VKMR_PACK_STREAM=1" "VKMR_PACK_STREAM=-1"; do
  env $v VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | python3 -c "
import sys,re
t={}
for l in sys.stdin:
    m=re.match(r'\[timing\] (.+): ([0-9.e+-]+) ms',l)
    if m: t[m.group(1).strip()]=float(m.group(2))
    m=re.search(r'=> ([0-9a-f]{8}).* in ([0-9.]+)\$',l)
    if m and 'computed root' in l: t['printed']=float(m.group(2)); t['root']=m.group(1)
print('%-22s root %s printed %6.1f  pass 1 %5.1f  pass 2 %5.1f  pipeline-full wait %5.1f' % ('$v', t.get('root'), t['printed'], t['pack pass 1 (index the lines, fork-join)'], t['pack pass 2 (copy the lines, fork-join)'], t['wait for the oldest mapping (pipeline full)']))"
done; done
} > gpurun_out/r03/tuner_ab.txt 2>&1
python3 - <<'PY'
import re,statistics
d={}
for l in open('gpurun_out/r03/tuner_ab.txt'):
    m=re.match(r'(VKMR_PACK_STREAM=-?\d)\s+root (\w+) printed\s+([0-9.]+)\s+pass 1\s+([0-9.]+)\s+pass 2\s+([0-9.]+)',l)
    if m: d.setdefault(m.group(1),[]).append((float(m.group(3)),float(m.group(5)),m.group(2)))
for k,v in d.items():
    p=[x[0] for x in v]; q=[x[1] for x in v]
    print(k,'printed median %.1f mean %.1f min %.1f max %.1f | pass 2 median %.1f'%(statistics.median(p),statistics.mean(p),min(p),max(p),statistics.median(q)), set(x[2] for x in v))
PY
