#!/bin/bash
# GPU box, repo root: parity suite, then the f32 bench lines of the 1024-point class.  usage: tools/gpu_wide.sh <tag> [pytest -k expr]
set -e
tag=${1:-wide}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
if [ -n "$2" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "$2" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
else
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
fi
tail -3 $out/pytest.log
for w in c4 c5; do
timeout -k 10 200 python3 bench.py --workload $w --precision ${PREC:-f32} --no-other --steps 20 --warmup 3 --no-cpu --no-host > $out/bench_$w.json 2> $out/bench_$w.err
python3 -c "
import json,sys
d=json.load(open('$out/bench_$w.json'))
print('$w', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'])
"
done
