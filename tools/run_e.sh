# round-2 measurement set (GPU box): parity suite, PMC passes, kernel stats, the bench line, end-to-end front end
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_r02_b.log 2>&1; tail -3 gpurun_out/pytest_r02_b.log
bash tools/pmc_profile.sh r02b > gpurun_out/pmc_r02b.log 2>&1
python3 tools/pmc_to_json.py gpurun_out/pmc_r02b gpurun_out/pmc_r02b.json > /dev/null
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --no-long-strings > $GRAFT_REPO_ROOT/gpurun_out/prof_r02b.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r02b.err )
cp gpurun_out/pmc_r02b.json profiles/pmc_latest.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_c.json 2> gpurun_out/bench_r02_c.err; echo "bench rc=$?"
# end to end from stdin: 2^25 strings (2.16 GB)
vk_merkle_roots_amd/bin/rndm 42 33554432 127 > /tmp/g25.txt 2>/dev/null
for i in 1 2 3; do vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>/dev/null | tail -1; done
for i in 1 2; do VKMR_PACK_THREADS=16 vk_merkle_roots_amd/bin/vkmr hip:0 < /tmp/g25.txt 2>/dev/null | tail -1; done
for i in 1 2; do cat /tmp/g25.txt | vk_merkle_roots_amd/bin/vkmr hip:0 2>/dev/null | tail -1; done
vk_merkle_roots_amd/bin/vkmr CPU < /tmp/g3.txt 2>/dev/null | tail -1
