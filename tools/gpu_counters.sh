#!/bin/bash
# GPU box, repo root: two SQ counter passes of one bench workload.  usage: tools/gpu_counters.sh <tag> <workload> [bench args]
tag=$1; w=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $out/sq_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu --no-host "$@" > $out/sq_$w.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq2_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu --no-host "$@" > $out/sq2_$w.log 2>&1 || exit 1
grep -o '"kernel_ms": [0-9.]*' $out/sq_$w.log
