#!/bin/bash
# round 4, call A: the new control structure (row ranges of the decode step, paced launch loop, kept sampler-step graph, overflow guards):
# parity tests, the whole -m gpu suite, the in-situ A/B of the row-range settings on ONE handle, kernel traces of two settings, a bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_r4.py -x -q -m gpu -s -p no:cacheprovider > $OUT/r4_tests.log 2>&1; echo "r4 tests rc=$?"
grep -E "passed|failed|Error|assert|\[guard\]" $OUT/r4_tests.log | tail -20
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?"
tail -5 $OUT/pytest_gpu.log
V="1,0,0;2,0,0;2,1,0;2,0,1;4,0,0;4,1,0;4,0,1;1,0,0,1;1,0,0"
timeout 600 python scripts/ab_stage.py ar --reps 3 --ar-variants "$V" > $OUT/ab_r4a.txt 2>&1; echo "ab rc=$?"
grep "^ab " $OUT/ab_r4a.txt
for v in "1,0,0" "2,0,0" "2,1,0" "4,1,0"; do
  tag=$(echo $v | tr ',' '_')
  rm -rf $OUT/trace_$tag
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$OUT/trace_$tag -o t -- python $OLDPWD/scripts/ab_stage.py ar --reps 1 --mel-tokens 24 --ar-variants "$v" > $OLDPWD/$OUT/trace_$tag.log 2>&1)
  f=$(find $OUT/trace_$tag -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/trace_overlap.py $f "ranges,graph,stagger=$v" | tee -a $OUT/trace_overlap_r4a.txt
  find $OUT/trace_$tag -type f -size +6M -delete
done
timeout 900 python bench.py > $OUT/bench_r4a.log 2> $OUT/bench_r4a.err; echo "bench rc=$?"
tail -1 $OUT/bench_r4a.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','dtype')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()}); r=d['roofline']; print({k:r[k] for k in ('kernel','frac','avg_launch_us')})"
exit 0
