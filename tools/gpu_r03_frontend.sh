cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
nproc; lscpu | grep -E "Model name|^CPU\(s\)|L2|L3"
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
echo "# the packer alone (tools/pack_bench): mapped file, then resident copy"
tools/pack_bench /tmp/g25.txt 16 32 0
tools/pack_bench /tmp/g25.txt 16 32 1
tools/pack_bench /tmp/g25.txt 8 32 0
tools/pack_bench /tmp/g25.txt 32 32 0
tools/pack_bench /tmp/g25.txt 1 32 1
echo "# VKMR_TIMING=1 vkmr hip:0 < file (2^25 strings)"
for i in 1 2 3; do VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | grep -v "^Initializing"; echo; done
for v in "VKMR_INPUT_SPAN_MB=8" "VKMR_INPUT_SPAN_MB=128" "VKMR_PACK_THREADS=8" "VKMR_PACK_THREADS=32" "VKMR_BATCH_MB=128" "VKMR_BATCH_MB=32"; do
  echo "# $v"
  for i in 1 2; do env $v VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | grep -E "computed root|pass|wait|drain|dispatch" ; echo; done
done
} > gpurun_out/r03/frontend.txt 2>&1
cat gpurun_out/r03/frontend.txt
