#!/bin/bash
# round 4, call D: plain slab stores - failure-rate probe, the round-4 tests (ranges, kept graphs, guards, two-thread stress), the whole -m gpu suite,
# the in-situ A/B of the row ranges at the benchmark shape, the stream-concurrency probe, a bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
RUNS=40 timeout 600 python scripts/debug_ranges.py 2>&1 | grep "^rate" | tee $OUT/debug_rate_plain.txt
timeout 900 python -m pytest tests/test_gpu_r4.py -q -m gpu -s -p no:cacheprovider > $OUT/r4_tests.log 2>&1; echo "r4 tests rc=$?"
grep -E "passed|failed|Error|^E  |\[guard\]|^FAILED" $OUT/r4_tests.log | tail -30
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?"
tail -8 $OUT/pytest_gpu.log
timeout 600 python scripts/ab_stage.py ar diff --reps 3 --ar-variants "1;2;4;1,1;1,2;1,12;1" > $OUT/ab_r4d.txt 2>&1; echo "ab rc=$?"
grep "^ab " $OUT/ab_r4d.txt
timeout 900 python bench.py > $OUT/bench_r4d.log 2> $OUT/bench_r4d.err; echo "bench rc=$?"
tail -1 $OUT/bench_r4d.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','dtype')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()}); r=d['roofline']; print({k:r[k] for k in ('kernel','frac','avg_launch_us')})"
exit 0
