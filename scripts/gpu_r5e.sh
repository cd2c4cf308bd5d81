#!/bin/bash
# round 5, call E: UnivNet audio-rate kernels on the f32 matrix cores (op tests, stage tests, A/B), flash32 with the bias as accumulator input (A/B),
# vocoder overflow guard, the whole -m gpu suite, a bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -s -p no:cacheprovider -k "univnet or flash" > $OUT/r5e_ops.log 2>&1; echo "op tests rc=$?"
grep -E "passed|failed|Error|assert|conv1d|lvc|convt" $OUT/r5e_ops.log | tail -20
timeout 300 python -m pytest tests/test_gpu_r5.py -q -m gpu -s -p no:cacheprovider -k "vocoder" > $OUT/r5e_guard.log 2>&1; echo "guard test rc=$?"; grep -E "passed|failed|guard|Error" $OUT/r5e_guard.log | tail -5
timeout 300 python scripts/ab_stage.py voc --dtype fp16 --voc-variants "1;0;1;0" > $OUT/ab_r5e_voc.txt 2>&1; echo "ab voc rc=$?"; grep "^ab " $OUT/ab_r5e_voc.txt; tail -2 $OUT/ab_r5e_voc.txt
timeout 400 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --flash-variants "1;0;1;0" > $OUT/ab_r5e.txt 2>&1; echo "ab diff rc=$?"; grep "^ab " $OUT/ab_r5e.txt
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?"; tail -6 $OUT/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r5e.log 2> $OUT/bench_r5e.err; echo "bench rc=$?"
tail -1 $OUT/bench_r5e.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','dtype')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()}); r=d['roofline']; print({k:r[k] for k in ('kernel','frac','avg_launch_us')}); print(r['stages']['univnet'])
for k in d['kernel_breakdown_ms']: print(k)"
exit 0
