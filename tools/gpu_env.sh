#!/bin/bash
# GPU box, repo root: one bench line per set of plan-heuristic overrides (bench.py --tune).
# usage: tools/gpu_env.sh <workload> "key=val,key=val" ["key=val" ...]   ("-" = no override)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MFCC_AMD_LIB=${MFCC_AMD_LIB:-$PWD/mfcc_amd/libmfcc_amd_dev.so}
w=$1; shift
for e in "$@"; do
t="debug=1"; [ "$e" != "-" ] && t="debug=1,$e"
timeout -k 10 200 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu --no-host --tune "$t" > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python3 -c "
import json
d=json.load(open('/tmp/b.json'))
print('$w [$e]', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'])
"; grep "mfcc plan" /tmp/b.err | grep -v "mel rounds\|matrix-core" | head -2
done
