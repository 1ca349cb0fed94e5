#!/bin/bash
# GPU box, repo root: one bench line per workload (and a two-rank, one-device rehearsal of the
# sharded one over gloo).  usage: tools/gpu_bench_all.sh <tag>
tag=${1:-benchall}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
for w in c1 c3 c4 c5; do
  timeout -k 10 400 python3 bench.py --workload $w --steps 20 --warmup 3 > $out/bench_$w.json 2> $out/bench_$w.err || { tail -5 $out/bench_$w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$out/bench_$w.json'))
print('$w', 'frames/s %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'], 'pass frac %.4f' % d['roofline']['whole_pass']['frac'], 'cpu %.4g (1 thread %.4g)' % (d['cpu_baseline']['value'], d['cpu_baseline']['one_thread']))
"
done
timeout -k 10 400 python3 bench.py > $out/bench_c2.json 2> $out/bench_c2.err || { tail -5 $out/bench_c2.err; exit 1; }
cut -c1-1500 $out/bench_c2.json
MFCC_BENCH_ONE_DEVICE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload c5 --utts 30 --steps 5 --warmup 2 > $out/bench_c5_2rank.json 2> $out/bench_c5_2rank.err || { tail -5 $out/bench_c5_2rank.err; exit 1; }
cut -c1-600 $out/bench_c5_2rank.json
