cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
lscpu | grep -i numa
for d in /sys/bus/pci/devices/*; do if [ -e $d/vendor ] && grep -q 0x1002 $d/vendor && [ -e $d/class ] && grep -q "^0x12\|^0x03" $d/class; then echo "$d class $(cat $d/class) numa_node $(cat $d/numa_node) local_cpulist $(cat $d/local_cpulist)"; fi; done
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
echo "node0 $N0"; echo "node1 $N1"
for round in 1 2 3 4; do
for c in "$N0" "$N1"; do
  [ -z "$c" ] && continue
  echo "taskset -c $c"
  VKMR_TIMING=1 taskset -c $c vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | grep -E "computed|pass 1|pass 2|two copies"
done; done
} > gpurun_out/r03/numa.txt 2>&1
cat gpurun_out/r03/numa.txt
