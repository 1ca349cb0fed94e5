#!/bin/bash
# GPU box, repo root: one double-precision bench line per workload.  usage: tools/gpu_f64_all.sh <tag> [workloads...]
tag=${1:-f64all}; shift
ws=${@:-c2 c3 c4 c1 c5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
for w in $ws; do
  timeout -k 10 300 python3 bench.py --workload $w --precision f64 --steps 10 --warmup 2 --no-cpu --no-host --tune debug=1 > $out/bench_$w.json 2> $out/bench_$w.err || { tail -5 $out/bench_$w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$out/bench_$w.json'))
print('$w f64', 'frames/s %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'])
"
  grep "mfcc plan" $out/bench_$w.err | grep -v "mel rounds" | head -3
done
