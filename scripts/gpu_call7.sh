#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?" | tee -a $OUT/summary.txt
grep -h "\[parity\]" $OUT/pytest_gpu.log > $OUT/parity.txt
tail -3 $OUT/pytest_gpu.log; grep -E "^FAILED|^ERROR|Error" $OUT/pytest_gpu.log | head
for v in base default base default; do
  if [ $v = default ]; then unset TORTOISE_MI355X_LIB; else export TORTOISE_MI355X_LIB=$PWD/tortoise_tts_amd/lib/libtt_$v.so; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench7_$v.log 2> $OUT/bench7_$v.err; echo "bench $v rc=$?" | tee -a $OUT/summary.txt
  python - <<PY
import json
d=json.loads(open('gpurun_out/bench7_$v.log').read().strip().splitlines()[-1])
print('$v', {k:round(v,4) for k,v in d['stages_s_per_step'].items()})
for r in d['kernel_breakdown_ms'][:14]: print('   %-36s %6d %9.3f %8.2f' % (r['kernel'], r['launches'], r['total_ms'], r['avg_us']))
PY
done
