cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1
for round in 1 2 3 4 5 6; do
  VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | grep -E "computed|timing"; echo
done
} > gpurun_out/r03/frontend5.txt 2>&1
grep -E "computed|enumerated|constructed|root printed|copies|index|pass 2|launch|drain" gpurun_out/r03/frontend5.txt
