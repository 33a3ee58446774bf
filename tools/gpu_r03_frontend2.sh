cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
echo "# VKMR_TIMING=1 vkmr hip:0 < file (2^25 strings), 6 runs"
for i in 1 2 3 4 5 6; do VKMR_TIMING=1 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | grep -E "computed root|timing"; echo; done
echo "# hip api stats of one run (rocprofv3 --hip-trace --stats)"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/fe_trace -- $GRAFT_REPO_ROOT/vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>&1 | grep "computed root"; find /tmp/fe_trace -name '*hip_api_stats.csv' -exec head -25 {} \; )
} > gpurun_out/r03/frontend2.txt 2>&1
cat gpurun_out/r03/frontend2.txt
