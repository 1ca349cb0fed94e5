#!/bin/bash
# GPU box, repo root: the c1 (one example file) bench line per finishing path, both precisions.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for prec in f32 f64; do
for e in "finish=3" "finish=1" "-"; do
t="debug=1"; [ "$e" != "-" ] && t="debug=1,$e"
timeout -k 10 200 python3 bench.py --workload c1 --precision $prec --steps 200 --warmup 10 --no-cpu --no-host --no-other --tune "$t" > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python3 -c "
import json
d=json.load(open('/tmp/b.json'))
print('c1 $prec [$e]', 'ms/step %.4f' % d['ms_per_step'], 'kernel %.4f' % d['roofline']['kernel_ms'])
"; grep "deltas / z-norm" /tmp/b.err | head -1
done; done
