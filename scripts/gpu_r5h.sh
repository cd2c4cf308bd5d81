#!/bin/bash
# round 5, call H: V^T written as 8-byte row quads from unswapped MFMAs (QKV GEMM epilogue) against the 2-byte-store form: parity (diffusion / CLVP /
# AR prefill at full width), per-class kernel time of the two builds (bench roofline leg), stage A/B, CLVP stage time
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
NOVT=$PWD/tortoise_tts_amd/lib/libtortoise_mi355x_novt.so
timeout 400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_stages.py tests/test_gpu_parity_r3.py tests/test_gpu_f32.py -q -m gpu -p no:cacheprovider -x > $OUT/r5h_parity.log 2>&1; echo "parity rc=$?"; tail -3 $OUT/r5h_parity.log
for lib in "" "$NOVT"; do
  tag=$([ -z "$lib" ] && echo quads || echo bytes)
  TORTOISE_MI355X_LIB=$lib timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', {k:round(v,4) for k,v in d['stages_s_per_step'].items()})
for k in d['kernel_breakdown_ms']:
    if 'QkvHeads' in k['kernel'] or 'flash' in k['kernel']: print('$tag', k)"
done
: > $OUT/ab_r5h.txt
for rep in 1 2; do
  timeout 200 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --tag quads >> $OUT/ab_r5h.txt 2>&1
  TORTOISE_MI355X_LIB=$NOVT timeout 200 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --tag bytes >> $OUT/ab_r5h.txt 2>&1
done
grep "^ab " $OUT/ab_r5h.txt
exit 0
