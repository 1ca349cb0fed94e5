#!/bin/bash
# GPU box, repo root: the parity tests that reach the double-precision pipelined kernel, then its bench lines.
# usage: tools/gpu_pipe64_check.sh <tag> [workloads]
tag=${1:-p64}; shift
ws=${@:-c2 c4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread \
  -k "named_configs or lean_kernel_modes or stereo_2048 or golden or zcr_is_exact or sharded_recording or ragged_batch" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for w in $ws; do
  timeout -k 10 300 python3 bench.py --workload $w --precision f64 --steps 10 --warmup 2 --no-cpu --no-host --tune debug=1 > $out/bench_$w.json 2> $out/bench_$w.err || { tail -5 $out/bench_$w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$out/bench_$w.json'))
print('$w f64', 'frames/s %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'])
"
  grep "mfcc plan" $out/bench_$w.err | grep -v "mel rounds" | head -3
done
