#!/bin/bash
# GPU box, repo root: parity suite, then bench lines and SQ counter passes of the c2 and c4 workloads.
# usage: tools/gpu_baseline.sh <tag>   (results under gpurun_out/<tag>/)
set -e
tag=${1:-base}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 > $out/bench_c2.json 2> $out/bench_c2.err
timeout -k 10 200 python3 bench.py --workload c4 --steps 10 --warmup 3 --no-cpu --no-host > $out/bench_c4.json 2> $out/bench_c4.err
timeout -k 10 200 python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu --no-host > $out/bench_c3.json 2> $out/bench_c3.err
for w in c2 c4; do
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $out/sq_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu --no-host > $out/sq_$w.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq2_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu --no-host > $out/sq2_$w.log 2>&1
done
cat $out/bench_c2.json $out/bench_c4.json $out/bench_c3.json | cut -c1-400
