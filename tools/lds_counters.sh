#!/bin/bash
# LDS / VALU counters of one bench configuration (GPU box, repo root): tools/lds_counters.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ldsc
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/ldsc/$1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/ldsc/$1.log 2>&1
echo "$1: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ldsc/$1.log)"
