#!/bin/bash
# round 5, call A: the five-launch decode step (EPI_RESID ticket fold + folded LayerNorm): operator + engine tests, the whole -m gpu suite,
# the in-situ A/B five- vs seven-launch step on ONE handle, the two-thread stress test, a bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_r5.py -x -q -m gpu -s -p no:cacheprovider > $OUT/r5_tests.log 2>&1; echo "r5 tests rc=$?"
grep -E "passed|failed|Error|assert|rel_l2" $OUT/r5_tests.log | tail -60
timeout 600 python scripts/ab_stage.py ar --reps 3 --ar-variants "1;0;1;0" > $OUT/ab_r5a.txt 2>&1; echo "ab rc=$?"
grep "^ab " $OUT/ab_r5a.txt; tail -3 $OUT/ab_r5a.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?"
tail -15 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_r5a.log 2> $OUT/bench_r5a.err; echo "bench rc=$?"
tail -1 $OUT/bench_r5a.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','dtype')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()}); r=d['roofline']; print({k:r[k] for k in ('kernel','frac','avg_launch_us')})
for k in d['kernel_breakdown_ms'][:14]: print(k)"
exit 0
