#!/bin/bash
# GPU box: end-to-end timings of the C entry points (host buffers / files in and out) into gpurun_out/entry_points.jsonl
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/entry_points.jsonl; : > $out
for p in f64 f32; do
  timeout -k 10 300 python3 tools/bench_recording.py --minutes 60 --precision $p --reps 3 >> $out 2> gpurun_out/entry_rec_$p.err || { tail -3 gpurun_out/entry_rec_$p.err; exit 1; }
done
for p in f64 f32; do
  timeout -k 10 300 python3 tools/bench_list.py --preset c2 --files 1000 --save b --precision $p --dir /tmp/bench_list_tmp >> $out 2> gpurun_out/entry_list_$p.err || { tail -3 gpurun_out/entry_list_$p.err; exit 1; }
done
timeout -k 10 300 python3 tools/bench_list.py --preset c2 --files 1000 --save f --precision f32 --dir /tmp/bench_list_tmp >> $out 2> gpurun_out/entry_list_f.err || { tail -3 gpurun_out/entry_list_f.err; exit 1; }
grep -c entry $out
