#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 600 python scripts/debug_ranges.py > $OUT/debug_ranges.txt 2>&1; echo "debug rc=$?"; cat $OUT/debug_ranges.txt | tail -20
timeout 600 python scripts/kbench.py streams > $OUT/kbench_streams.txt 2>&1; echo "kbench rc=$?"; cat $OUT/kbench_streams.txt | tail -8
timeout 900 python -m pytest tests/test_gpu_r4.py -q -m gpu -s -p no:cacheprovider > $OUT/r4_tests.log 2>&1; echo "r4 tests rc=$?"
grep -E "passed|failed|Error|^E  |\[guard\]|^FAILED" $OUT/r4_tests.log | tail -30
exit 0
