cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
for round in 1 2; do
for m in 0 2 3 1; do tools/pack_bench /tmp/g25.txt 16 32 $m | head -3; done
done
} > gpurun_out/r03/populate.txt 2>&1
cat gpurun_out/r03/populate.txt
