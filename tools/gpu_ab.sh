#!/bin/bash
# GPU box, repo root: A/B of two builds of the library on one box.  usage: tools/gpu_ab.sh <workload> <precision> [libA] [libB]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=${1:-c4}; p=${2:-f32}; A=${3:-mfcc_amd/libmfcc_amd.so}; B=${4:-mfcc_amd/libmfcc_amd_alt.so}
for i in 1 2 3; do
for L in $A $B; do
MFCC_AMD_LIB=$PWD/$L timeout -k 10 200 python3 bench.py --workload $w --precision $p --no-other --steps 20 --warmup 3 --no-cpu --no-host > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
python3 -c "
import json
d=json.load(open('gpurun_out/ab.json'))
print('$L', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'])
"
done
done
