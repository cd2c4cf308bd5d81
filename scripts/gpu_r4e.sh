#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
timeout 900 python -m pytest tests/test_gpu_r4.py -q -m gpu -s -p no:cacheprovider -k "overlapped or kept or two_engines" > $OUT/r4_tests.log 2>&1; echo "r4 tests rc=$?"
grep -E "passed|failed|Error|^E  |^FAILED" $OUT/r4_tests.log | tail -20
: > $OUT/ab_r4e.txt
for v in 0 1 0 1; do TT_DIFF_OVERLAP_PREPASS=$v AB_TAG=overlap$v timeout 300 python scripts/ab_stage.py diff --reps 3 --dtype fp16 2>&1 | grep "^ab " | tee -a $OUT/ab_r4e.txt; done
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?"
tail -4 $OUT/pytest_gpu.log
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_r4e.log 2> $OUT/bench_r4e.err; echo "bench rc=$?"
tail -1 $OUT/bench_r4e.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','dtype')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()})"
exit 0
