cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1
for round in 1 2 3; do
for w in 256 65536 1048576 8388608 41943040; do
  echo "VKMR_WARM_BYTES=$w"
  VKMR_WARM_BYTES=$w VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | grep -E "computed|enumerated|constructed|two copies"
done; done
} > gpurun_out/r03/frontend6.txt 2>&1
cat gpurun_out/r03/frontend6.txt
