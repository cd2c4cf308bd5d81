#!/bin/bash
# round 5, call B: flash32 (32-query waves on v_mfma_f32_32x32x16): operator tests vs torch, the in-situ A/B on the diffusion stage, CLVP timing,
# the -m gpu suite, a bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -s -p no:cacheprovider -k flash > $OUT/r5b_flash_tests.log 2>&1; echo "flash tests rc=$?"
grep -E "passed|failed|Error|assert|flash" $OUT/r5b_flash_tests.log | tail -40
timeout 900 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --flash-variants "1;0;1;0" > $OUT/ab_r5b.txt 2>&1; echo "ab rc=$?"
grep "^ab " $OUT/ab_r5b.txt; tail -3 $OUT/ab_r5b.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?"
tail -8 $OUT/pytest_gpu.log
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_r5b.log 2> $OUT/bench_r5b.err; echo "bench rc=$?"
tail -1 $OUT/bench_r5b.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','dtype')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()}); r=d['roofline']; print({k:r[k] for k in ('kernel','frac','avg_launch_us')})
for k in d['kernel_breakdown_ms'][:16]: print(k)"
exit 0
