#!/bin/bash
# Phase ablation with the diagnostic library (results are wrong by design, only the time matters):
# GPU box, repo root.  usage: tools/ablate.sh <workload> <mask> [<mask> ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MFCC_AMD_LIB=$PWD/mfcc_amd/libmfcc_amd_skip.so
w=$1; shift
for m in "$@"; do
  export MFCC_SKIP=$m
  timeout -k 10 120 python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu --no-host > /tmp/abl.json 2>/tmp/abl.err || { cat /tmp/abl.err | tail -5; exit 1; }
  echo "mask $m: $(grep -o '"kernel_ms": [0-9.]*' /tmp/abl.json)"
done
