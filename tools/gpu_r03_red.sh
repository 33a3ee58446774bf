cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > gpurun_out/r03/pytest_tail.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_tail.log
for r in 1 2; do python3 tools/reduce_probe.py 26 20; done > gpurun_out/r03/reduce_probe3.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/redprof && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/redprof -- python3 $GRAFT_REPO_ROOT/tools/reduce_probe.py 26 20 > /dev/null 2>&1; python3 -c "
import csv, glob
for f in glob.glob('/tmp/redprof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)): print(r['Name'][:28], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3, 'max', float(r['MaxNs'])/1e3)" ) >> gpurun_out/r03/reduce_probe3.txt 2>&1
cat gpurun_out/r03/reduce_probe3.txt
for r in 1 2; do python3 tools/long_strings_probe.py 21 4096; done
