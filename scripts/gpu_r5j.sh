#!/bin/bash
# round 5, call J: eight-phase GEMM tile - epilogue depth / lgkmcnt knobs (variant libraries), then in situ: CLVP + pre-pass users (default bench line) and the
# long-form reading workload with the tile switched on / off (TT_GEMM_VARIANT)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
: > $OUT/ab_r5j_gemm.txt
for v in "" _ed3 _nolgkm; do
  echo "--- library libtortoise_mi355x$v.so" >> $OUT/ab_r5j_gemm.txt
  TORTOISE_MI355X_LIB=$PWD/tortoise_tts_amd/lib/libtortoise_mi355x$v.so timeout 200 python scripts/ab_gemm_p8.py --quick >> $OUT/ab_r5j_gemm.txt 2>&1
done
grep "^ab \|^---\|rror" $OUT/ab_r5j_gemm.txt
for v in 1 0 1 0; do
  TT_GEMM_VARIANT=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench gemm_variant=$v', round(d['ms_per_step'],1), {k:round(x,4) for k,x in d['stages_s_per_step'].items()})"
done
for v in 1 0; do
  TT_GEMM_VARIANT=$v timeout 300 python bench.py --workload read --steps 1 --warmup 1 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('read gemm_variant=$v', round(d['value'],2), round(d['ms_per_step'],1), {k:round(x,3) for k,x in d.get('stages_s_per_step',{}).items()})"
done
exit 0
