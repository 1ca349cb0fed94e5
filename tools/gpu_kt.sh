#!/bin/bash
# GPU box, repo root: per-kernel times (rocprofv3 kernel trace) of one bench workload.  usage: tools/gpu_kt.sh <tag> <workload> [bench args]
tag=$1; w=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$w -- python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu --no-host --no-other "$@" > $out/kt_$w.log 2>&1 || { tail -5 $out/kt_$w.log; exit 1; }
cp $out/kt_$w/*/*_kernel_stats.csv $out/kernel_stats_$w.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats_$w.csv")))
for r in rows[:8]:
    print("%-90s calls %5s avg %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
