#!/bin/bash
# round 5, call I: the eight-phase 256 x 256 GEMM tile (csrc/gemm_p8.h) - bit-identity against the 16-wave tile, large-M op tests, in-situ timing
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_r5.py tests/test_gpu_ops.py -q -m gpu -s -p no:cacheprovider -k "eight_phase or gemm_large" > $OUT/r5i_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error|assert|eight-phase" $OUT/r5i_tests.log | tail -8
timeout 300 python scripts/ab_gemm_p8.py > $OUT/ab_r5i_gemm.txt 2>&1; echo "ab rc=$?"; grep "^ab \|Error\|error" $OUT/ab_r5i_gemm.txt | head -20
exit 0
