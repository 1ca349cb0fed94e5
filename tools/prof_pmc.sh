#!/bin/bash
# usage: tools/prof_pmc.sh <outdir under gpurun_out> [bench args...]; run on the GPU box from the repo root
set -e
out=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu "$@" > $out/kt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/pmc1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu "$@" > $out/pmc1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $out/pmc2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu "$@" > $out/pmc2.log 2>&1
