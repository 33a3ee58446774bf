cd $GRAFT_REPO_ROOT
bash tools/fetch_calibrate.sh > gpurun_out/fetch_calibrate.log 2>&1
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -q -k "512_mib" > gpurun_out/pytest_bigstring.log 2>&1; tail -5 gpurun_out/pytest_bigstring.log
timeout -k 10 300 python bench.py --gpus 2 --rehearse-gloo --steps 3 --warmup 1 > gpurun_out/bench_r02_gloo2.json 2> gpurun_out/bench_r02_gloo2.err; echo "rc=$?"; head -c 300 gpurun_out/bench_r02_gloo2.json
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline --no-pipeline --no-long-strings > gpurun_out/bench_r02_dist1.json 2> gpurun_out/bench_r02_dist1.err; echo "rc=$?"; head -c 200 gpurun_out/bench_r02_dist1.json
