#!/bin/bash
# GPU call: full -m gpu suite (parity log), kernel experiments, default bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
rocminfo 2>/dev/null | grep -E "gfx|Marketing" | head -4 > $OUT/device.txt; nproc >> $OUT/device.txt
timeout 1500 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?" | tee -a $OUT/summary.txt
grep -h "\[parity\]" $OUT/pytest_gpu.log > $OUT/parity.txt
timeout 600 python scripts/kbench.py > $OUT/kbench.log 2>&1; echo "kbench rc=$?" | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 2 --warmup 1 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log; grep -E "FAILED|Error" $OUT/pytest_gpu.log | head -20; tail -5 $OUT/kbench.log; tail -c 1500 $OUT/bench.log
