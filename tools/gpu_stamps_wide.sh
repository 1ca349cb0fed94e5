#!/bin/bash
# GPU box, repo root: phase table of the 2048-point class kernels (diagnostic stamps library).  usage: tools/gpu_stamps_wide.sh <f32|f64>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MFCC_AMD_LIB=$PWD/mfcc_amd/libmfcc_amd_stamps.so MFCC_STAMPS_DUMP=1
prec=${1:-f32}
if [ "$prec" = f32 ]; then set -- "0x1 1:expansion" "0x2 2:table loads" "0x4 3:FFT" "0x8 4:split+weights" "0x10 7:sub-bands" "0x80 5:products" "0x20 8:part sums" "0x100 6:joins+log" "0x40 9:cepstra" "0x200 0:advance" "0x1 6:whole round";
else set -- "0x1 1:expansion" "0x2 3:FFT" "0x8 4:split+power" "0x10 5:next bytes asked + sub-bands" "0x20 6:mel" "0x40 7:next samples" "0x80 8:stores" "0x100 9:cepstra" "0x200 0:advance" "0x1 6:whole round"; fi
for iv in "$@"; do
  spec=${iv%%:*}; name=${iv#*:}
  set -- $spec
  export MFCC_STAMP_BEGIN=$1 MFCC_STAMP_END=$2
  echo -n "$name: "
  timeout -k 10 120 python3 bench.py --workload c4 --precision $prec --steps 2 --warmup 1 --no-cpu --no-host --no-other 2>&1 | grep "mfcc stamps: interval" | tail -1
done
