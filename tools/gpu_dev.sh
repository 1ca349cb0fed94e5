#!/bin/bash
# GPU box, repo root: bench lines with the quick-turnaround library.  usage: tools/gpu_dev.sh <workload> [...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MFCC_AMD_LIB=$PWD/mfcc_amd/libmfcc_amd_dev.so
for w in "$@"; do
timeout -k 10 200 python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu --no-host > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python3 -c "
import json
d=json.load(open('/tmp/b.json'))
print('$w', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'])
"
done
