#!/bin/bash
# Collects the rocprofv3 evidence of the default bench workload into gpurun_out/profiles_r1/
# (run on the GPU box from the repo root; copy the summaries into profiles/ afterwards).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_r1
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 50 --warmup 5 --no-cpu --no-host > $out/kt_bench.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $out/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $out/write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d $out/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $out/sq.log 2>&1
cp $out/kt/*/*_kernel_stats.csv $out/kernel_stats.csv
grep -h metric $out/kt_bench.log > $out/bench_under_rocprof.json || true
