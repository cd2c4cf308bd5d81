#!/bin/bash
# Runs on the MI355X box (via gpurun).  Everything is logged under gpurun_out/ so a cut-off call still leaves evidence.
# usage: scripts/gpu_round.sh [phase ...]   phases: ops stages smoke bench prof
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
PHASES="${@:-ops stages smoke}"
rocminfo 2>/dev/null | grep -E "gfx|Marketing" | head -4 > $OUT/device.txt
nproc >> $OUT/device.txt
for ph in $PHASES; do
  case $ph in
    abi)    timeout 300 python -m pytest tests/test_abi.py -x -q > $OUT/abi.log 2>&1; echo "abi rc=$?" | tee -a $OUT/summary.txt ;;
    ops)    timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -s -p no:cacheprovider > $OUT/ops.log 2>&1; echo "ops rc=$?" | tee -a $OUT/summary.txt ;;
    stages) timeout 1200 python -m pytest tests/test_gpu_stages.py -q -m gpu -s -p no:cacheprovider > $OUT/stages.log 2>&1; echo "stages rc=$?" | tee -a $OUT/summary.txt ;;
    full)   timeout 1500 python -m pytest tests/test_gpu_full.py -q -m gpu -s -p no:cacheprovider > $OUT/full.log 2>&1; echo "full rc=$?" | tee -a $OUT/summary.txt ;;
    gpu)    timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?" | tee -a $OUT/summary.txt ;;
    smoke)  timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt ;;
    bench)  timeout 1500 python bench.py ${BENCH_ARGS:-} > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt; tail -2 $OUT/bench.log ;;
    prof)   rm -rf $OUT/prof; (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o prof -- python $OLDPWD/bench.py ${PROF_ARGS:---steps 1 --warmup 1 --no-cpu-baseline} > $OLDPWD/$OUT/prof.log 2>&1); echo "prof rc=$?" | tee -a $OUT/summary.txt
            find $OUT/prof -type f -size +8M -delete; find $OUT/prof -type f | head -20; tail -3 $OUT/prof.log ;;
  esac
done
grep -h "\[parity\]" $OUT/*.log 2>/dev/null | sed "s/^\.*//" > $OUT/parity.txt
tail -25 $OUT/summary.txt
for f in ops stages full smoke; do [ -f $OUT/$f.log ] && { echo "--- $f"; grep -E "passed|failed|error|Error|FAILED|assert" $OUT/$f.log | tail -30; }; done
exit 0
