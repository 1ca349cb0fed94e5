#!/bin/bash
# GPU box: f64 parity subset, then the f64 bench lines of the 256-point class workloads
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c2f64
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "named_configs or lean_kernel or every_fft or golden" > gpurun_out/c2f64/pytest.log 2>&1 || { tail -30 gpurun_out/c2f64/pytest.log; exit 1; }
tail -1 gpurun_out/c2f64/pytest.log
for w in c2 c3; do
timeout -k 10 200 python3 bench.py --workload $w --precision f64 --no-other --steps 20 --warmup 3 --no-cpu --no-host > gpurun_out/c2f64/bench_$w.json 2> gpurun_out/c2f64/bench_$w.err
python3 -c "
import json
d=json.load(open('gpurun_out/c2f64/bench_$w.json'))
print('$w', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'])
"
done
