#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -s -p no:cacheprovider -k "flash or groupnorm" > $OUT/pytest_flash.log 2>&1; echo "pytest-flash rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_flash.log; grep -E "^FAILED|^ERROR|rel_l2" $OUT/pytest_flash.log | head -40
for v in 9 1 2 auto; do
  if [ $v = auto ]; then unset TT_FLASH_NQ; else export TT_FLASH_NQ=$v; fi
  timeout 300 python scripts/kbench.py flash 2>&1 | grep -v amdgpu
done
unset TT_FLASH_NQ
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head
for v in base default; do
  if [ $v = default ]; then unset TORTOISE_MI355X_LIB; else export TORTOISE_MI355X_LIB=$PWD/tortoise_tts_amd/lib/libtt_$v.so; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench8_$v.log 2> $OUT/bench8_$v.err; echo "bench $v rc=$?" | tee -a $OUT/summary.txt
  python - <<PY
import json
d=json.loads(open('gpurun_out/bench8_$v.log').read().strip().splitlines()[-1])
print('$v', {k:round(v,4) for k,v in d['stages_s_per_step'].items()})
for r in d['kernel_breakdown_ms'][:12]: print('   %-36s %6d %9.3f %8.2f' % (r['kernel'], r['launches'], r['total_ms'], r['avg_us']))
PY
done
