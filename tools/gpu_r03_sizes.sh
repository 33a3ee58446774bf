cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "from_sizes or warm_up or golden" 2>&1 | tail -5
timeout -k 10 600 python -m pytest tests/test_frontend.py -m gpu -q -x 2>&1 | tail -5
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1
for round in 1 2 3 4 5; do
for v in "VKMR_SEND_METADATA=0" "VKMR_SEND_METADATA=1"; do
  echo "# $v"
  env $v VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | grep -E "computed|pipeline full|two copies|pass 1|pass 2"
done; done
} > gpurun_out/r03/sizes_ab.txt 2>&1
grep -E "^#|computed" gpurun_out/r03/sizes_ab.txt
