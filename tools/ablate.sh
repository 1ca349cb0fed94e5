#!/bin/bash
# Phase ablation with the diagnostic library: time + LDS counters per skip mask (GPU box, repo root).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MFCC_AMD_LIB=$PWD/mfcc_amd/libmfcc_amd_skip.so
rm -rf gpurun_out/abl; mkdir -p gpurun_out/abl
for m in "$@"; do
  export MFCC_SKIP=$m
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/abl/m$m -- python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/abl/m$m.log 2>&1
  echo "mask $m: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abl/m$m.log)"
done
