cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
E=$GRAFT_REPO_ROOT/build/ab
for lib in default $E/libc4lds.so $E/libc1lds.so; do
  if [ "$lib" = default ]; then unset VKMR_HIP_LIB; else export VKMR_HIP_LIB=$lib; fi
  echo "== $lib"; python3 tools/reduce_probe.py 26 20
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/redprof && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/redprof -- python3 $GRAFT_REPO_ROOT/tools/reduce_probe.py 26 20 > /dev/null 2>&1; python3 -c "
import csv, glob
for f in glob.glob('/tmp/redprof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'collapse' in r['Name'] or 'tail' in r['Name']: print(r['Name'][:28], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3, 'max', float(r['MaxNs'])/1e3)" )
done > gpurun_out/r03/collapse_placement.txt 2>&1
cat gpurun_out/r03/collapse_placement.txt
