#!/bin/bash
# GPU box: FETCH_SIZE of the calibration reads (tools/calib/fetch_calib.hip), KiB per launch and the factor to the known 1 GiB
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/calib; rm -rf $out; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- tools/calib/fetch_calib > $out/log 2>&1 || { tail -5 $out/log; exit 1; }
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$out/fetch/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='FETCH_SIZE': acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    kib=sum(v)/len(v)
    print('%-10s FETCH_SIZE %.0f KiB per launch  -> known bytes / counter bytes = %.3f'%(k,kib,(1<<30)/(kib*1024)))
PY
