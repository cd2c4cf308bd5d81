#!/bin/bash
# round 5, call G: flash32 with a 4-way key split (16 waves, 128-key tiles) for launches of <= one workgroup per CU; the 256 x 256 tile's skip read one
# strip ahead (A/B against a build without it: diffusion stage with the pre-pass serialised in front, and the long-form reading workload)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -s -p no:cacheprovider -k "flash or gemm_large" > $OUT/r5g_ops.log 2>&1; echo "op tests rc=$?"; grep -E "passed|failed|Error|assert" $OUT/r5g_ops.log | tail -5
timeout 500 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --flash-variants "1;2;0;1;2;0" > $OUT/ab_r5g_flash.txt 2>&1; echo "ab flash rc=$?"; grep "^ab " $OUT/ab_r5g_flash.txt
NOPF=$PWD/tortoise_tts_amd/lib/libtortoise_mi355x_nopf.so
: > $OUT/ab_r5g_epi.txt
for rep in 1 2; do
  TT_DIFF_OVERLAP_PREPASS=0 timeout 200 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --tag ahead >> $OUT/ab_r5g_epi.txt 2>&1
  TT_DIFF_OVERLAP_PREPASS=0 TORTOISE_MI355X_LIB=$NOPF timeout 200 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --tag at_use >> $OUT/ab_r5g_epi.txt 2>&1
done
grep "^ab " $OUT/ab_r5g_epi.txt
timeout 300 python bench.py --workload read --steps 1 --warmup 1 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('read ahead ', round(d['value'],2), round(d['ms_per_step'],1))"
TORTOISE_MI355X_LIB=$NOPF timeout 300 python bench.py --workload read --steps 1 --warmup 1 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('read at_use', round(d['value'],2), round(d['ms_per_step'],1))"
timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity_r3.py -q -m gpu -p no:cacheprovider > $OUT/r5g_parity.log 2>&1; echo "parity rc=$?"; tail -3 $OUT/r5g_parity.log
exit 0
