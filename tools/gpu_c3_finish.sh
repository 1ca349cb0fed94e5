#!/bin/bash
# GPU box, repo root: the c3 bench line with each finishing path (tuning key finish) and chunk size.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for e in "finish=1" "finish=3" "finish=2" "finish=2,streamRows=16" "finish=2,streamRows=64" "-"; do
t="debug=1"; [ "$e" != "-" ] && t="debug=1,$e"
timeout -k 10 200 python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu --no-host --no-other --tune "$t" $BENCH_EXTRA > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python3 -c "
import json
d=json.load(open('/tmp/b.json'))
print('c3 [$e]', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'])
"; grep "deltas / z-norm" /tmp/b.err | head -1
done
