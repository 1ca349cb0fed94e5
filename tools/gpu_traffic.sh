#!/bin/bash
# GPU box: kernel time and HBM byte counters of one workload / precision (quick look).  usage: tools/gpu_traffic.sh <w> <prec>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=${1:-c4}; p=${2:-f32}
out=gpurun_out/traffic_${w}_$p; rm -rf $out; mkdir -p $out
B="python3 bench.py --workload $w --precision $p --no-other --no-cpu --no-host $EXTRA"
timeout -k 10 200 $B --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err || exit 1
python3 -c "
import json
d=json.load(open('$out/bench.json'))
print('$w $p', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'])
"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B --steps 3 --warmup 1 > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B --steps 3 --warmup 1 > $out/write.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
def avg(d,c):
    tot=n=0
    for f in glob.glob('$out/'+d+'/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name']==c and ('k_wide1024' in r['Kernel_Name'] or 'k_pipe64' in r['Kernel_Name'] or 'k_lean256' in r['Kernel_Name']):
                tot+=float(r['Counter_Value']); n+=1
    return tot/max(n,1)
f=avg('fetch','FETCH_SIZE'); w=avg('write','WRITE_SIZE')
print('read GB %.2f write GB %.2f total %.2f'%(2*f*1024/1e9, w*1024/1e9, (2*f+w)*1024/1e9))
PY
