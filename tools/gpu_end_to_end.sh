cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_frontend.py -m gpu -q > gpurun_out/pytest_frontend.log 2>&1; tail -3 gpurun_out/pytest_frontend.log
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
for i in 1 2 3 4; do vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>/dev/null | tail -1; done
for i in 1 2; do VKMR_PACK_THREADS=8 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>/dev/null | tail -1; done
for i in 1 2; do cat /tmp/g25.txt | vk_merkle_roots_amd/bin/vkmr hip:0 2>/dev/null | tail -1; done
# wall clock of the whole process (HIP start-up and teardown included)
for i in 1 2 3; do /usr/bin/env bash -c 'S=$(date +%s.%N); vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt > /dev/null 2>&1; E=$(date +%s.%N); echo "process wall $(echo "$E - $S" | bc) s"'; done
vk_merkle_roots_amd/bin/rndm 42 4194304 4096 > /tmp/l22.txt 2>/dev/null; ls -la /tmp/l22.txt
for i in 1 2; do VKMR_VERBOSE=0 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/l22.txt 2>/dev/null | tail -1; done
