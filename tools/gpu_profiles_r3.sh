#!/bin/bash
# GPU box: the round's rocprofv3 evidence for the listed workload:precision pairs
# usage: tools/gpu_profiles_r3.sh c2:f32 c2:f64 ...
for wp in "$@"; do
  w=${wp%%:*}; p=${wp##*:}
  tools/collect_profiles_r3.sh $w $p > gpurun_out/prof_r3_${w}_$p.log 2>&1 || { echo "FAILED $wp"; tail -5 gpurun_out/prof_r3_${w}_$p.log; exit 1; }
  echo "ok $wp"
done
