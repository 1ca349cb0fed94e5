#!/bin/bash
# GPU box, repo root: the rocprofv3 evidence of one bench workload into gpurun_out/profiles_r2/<w>/
# (kernel trace + stats of a steady-state run, HBM byte counters in separate passes, two SQ counter
# passes).  Summaries are made afterwards with tools/profile_summaries.py.
# usage: tools/collect_profiles_r2.sh <workload> [steps]
w=${1:-c2}; steps=${2:-50}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_r2/$w
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --workload $w --steps $steps --warmup 5 --no-cpu --no-host > $out/kt_bench.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu --no-host > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu --no-host > $out/write.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $out/sq -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu --no-host > $out/sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq2 -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu --no-host > $out/sq2.log 2>&1 || exit 1
cp $out/kt/*/*_kernel_stats.csv $out/kernel_stats.csv
grep -h '"metric"' $out/kt_bench.log > $out/bench_under_rocprof.json || true
echo "$w done"
