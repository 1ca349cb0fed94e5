#!/bin/bash
# GPU box, repo root: the rocprofv3 evidence of one bench workload in one precision into
# gpurun_out/profiles_r3/<w>_<prec>/ (kernel trace + stats of a steady-state run, HBM byte counters
# in separate passes, two SQ counter passes; every pass a run of its own under its own limit).
# Summaries are made afterwards with tools/profile_summaries_r3.py.
# usage: tools/collect_profiles_r3.sh <workload> <f32|f64> [steps]
w=${1:-c2}; p=${2:-f32}; steps=${3:-50}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_r3/${w}_$p
rm -rf $out && mkdir -p $out
B="python3 bench.py --workload $w --precision $p --no-other --no-cpu --no-host"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B --steps $steps --warmup 5 > $out/kt_bench.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B --steps 3 --warmup 1 > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B --steps 3 --warmup 1 > $out/write.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $out/sq -- $B --steps 3 --warmup 1 > $out/sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq2 -- $B --steps 3 --warmup 1 > $out/sq2.log 2>&1 || exit 1
cp $out/kt/*/*_kernel_stats.csv $out/kernel_stats.csv
grep -h '"metric"' $out/kt_bench.log > $out/bench_under_rocprof.json || true
# keep what travels back small: the per-dispatch traces are not needed once the stats are made
rm -rf $out/kt
echo "$w $p done"
