cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
E=$GRAFT_REPO_ROOT/build/ab
for spec in "default:" "plain:$E/libplain.so" "exp_v5_window:$E/libexp.so:5" "exp_v4_direct:$E/libexp.so:4"; do
  label=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; var=${rest#*:}; [ "$var" = "$rest" ] && var=""
  ( [ -n "$lib" ] && export VKMR_HIP_LIB=$lib; [ -n "$var" ] && export VKMR_MAP_VARIANT=$var
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lf && timeout -k 10 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/lf -- python3 $GRAFT_REPO_ROOT/tools/long_strings_probe.py > /tmp/lf.log 2>&1
    python3 -c "
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/lf/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'map_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
d=[]
for f in glob.glob('/tmp/lf/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'map_kernel' in r['Kernel_Name']: d.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
m=lambda k: sum(acc[k])/max(1,len(acc[k]))
print('$label', 'FETCH_SIZE x2 GB per launch', round(m('FETCH_SIZE')*1024*2/1e9,3), 'TCC hit', round(m('TCC_HIT_sum')/1e6,1), 'M miss', round(m('TCC_MISS_sum')/1e6,1), 'M; kernel ms', round(sum(d)/max(1,len(d)),3), 'launches', len(d))" )
done > gpurun_out/r03/long_fetch.txt 2>&1
cat gpurun_out/r03/long_fetch.txt
