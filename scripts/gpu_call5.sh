#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
for v in default epi0 epi2 default; do
  if [ $v = default ]; then unset TORTOISE_MI355X_LIB; else export TORTOISE_MI355X_LIB=$PWD/tortoise_tts_amd/lib/libtt_$v.so; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench5_$v.log 2> $OUT/bench5_$v.err; echo "bench $v rc=$?" | tee -a $OUT/summary.txt
  python - <<PY
import json
d=json.loads(open('gpurun_out/bench5_$v.log').read().strip().splitlines()[-1])
print('$v', {k:round(v,4) for k,v in d['stages_s_per_step'].items()})
for r in d['kernel_breakdown_ms'][:9]: print('   ', r['kernel'], r['avg_us'])
PY
done
