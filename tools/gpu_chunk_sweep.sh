#!/bin/bash
# GPU box: kernel time of one workload / precision over the frames-per-comb tuning key.  usage: tools/gpu_chunk_sweep.sh <w> <prec> <c1> <c2> ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=$1; p=$2; shift 2
for c in "$@"; do
timeout -k 10 200 python3 bench.py --workload $w --precision $p --no-other --no-cpu --no-host --steps 20 --warmup 3 --tune leanChunk=$c > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { tail -3 gpurun_out/sweep.err; exit 1; }
python3 -c "
import json
d=json.load(open('gpurun_out/sweep.json'))
print('$w $p chunk $c', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'])
"
done
