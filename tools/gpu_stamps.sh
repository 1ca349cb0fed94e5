#!/bin/bash
# GPU box, repo root: cycles of each phase interval of the static-feature kernel (diagnostic library
# built by `make -C mfcc_amd/csrc stamps`).  usage: tools/gpu_stamps.sh <workload> ["begin_mask end" ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MFCC_AMD_LIB=$PWD/mfcc_amd/libmfcc_amd_stamps.so MFCC_STAMPS_DUMP=1
w=$1; shift
if [ $# -eq 0 ]; then set -- "0x8200 0" "0x1 1" "0x82 2" "0x4 3" "0x8 4" "0x10 5" "0x20 6" "0x80 8" "0x100 9" "0x8000 9"; fi
for iv in "$@"; do
  set -- $iv
  export MFCC_STAMP_BEGIN=$1 MFCC_STAMP_END=$2
  timeout -k 10 120 python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu --no-host 2>&1 | grep "mfcc stamps: interval" | tail -1
done
