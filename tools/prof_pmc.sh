#!/bin/bash
# usage: tools/prof_pmc.sh <outdir under gpurun_out> [bench args...]; run on the GPU box from the repo root
set -e
out=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu "$@" > $out/kt.log 2>&1
i=1
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
  "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32" ; do
  timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu "$@" > $out/pmc$i.log 2>&1
  i=$((i+1))
done
