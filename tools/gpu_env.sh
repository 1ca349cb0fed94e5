#!/bin/bash
# GPU box, repo root: one bench line per environment setting.  usage: tools/gpu_env.sh <workload> "VAR=val ..." ["VAR=val ..." ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MFCC_AMD_LIB=${MFCC_AMD_LIB:-$PWD/mfcc_amd/libmfcc_amd_dev.so}
w=$1; shift
for e in "$@"; do
( export $e; timeout -k 10 200 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu --no-host > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python3 -c "
import json
d=json.load(open('/tmp/b.json'))
print('$w [$e]', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'])
"; grep "mfcc plan" /tmp/b.err | head -3 )
done
