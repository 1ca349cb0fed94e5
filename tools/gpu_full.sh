#!/bin/bash
# GPU box, repo root: the whole GPU suite, then the randomised sweep with MFCC_FUZZ_SEEDS seeds (default 600).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/full
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/full/pytest.log 2>&1 || { tail -40 gpurun_out/full/pytest.log; exit 1; }
tail -2 gpurun_out/full/pytest.log
MFCC_FUZZ_SEEDS=${MFCC_FUZZ_SEEDS:-600} timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 --timeout-method thread -k random_configurations > gpurun_out/full/fuzz.log 2>&1
grep -E "^FAILED|passed|failed" gpurun_out/full/fuzz.log | tail -8
