#!/bin/bash
# GPU box, repo root: timing of several builds of the library on one box, interleaved.
# usage: tools/gpu_abn.sh <workload> <precision> <lib> [<lib> ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=$1; p=$2; shift 2
for i in 1 2 3; do
for L in "$@"; do
MFCC_AMD_LIB=$PWD/$L timeout -k 10 200 python3 bench.py --workload $w --precision $p --no-other --steps 20 --warmup 3 --no-cpu --no-host > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
python3 -c "
import json
d=json.load(open('gpurun_out/ab.json'))
print('$L', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'])
"
done
done
