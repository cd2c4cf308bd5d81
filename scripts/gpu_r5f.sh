#!/bin/bash
# round 5, call F: do the denoiser's kernel boundaries shrink when the f32 GEMM results are written THROUGH (sc1 + per-wave drain) instead of
# being left dirty for the boundary to flush?  In-situ A/B of two builds (product vs -DTT_WT_F32), the two-thread stress test and the
# full-width denoiser parity on the variant; then the whole -m gpu suite on the product build.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
WT=$PWD/tortoise_tts_amd/lib/libtortoise_mi355x_wt.so
: > $OUT/ab_r5f.txt
for rep in 1 2; do
  timeout 200 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --tag product >> $OUT/ab_r5f.txt 2>&1
  TORTOISE_MI355X_LIB=$WT timeout 200 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --tag writethru >> $OUT/ab_r5f.txt 2>&1
done
grep "^ab " $OUT/ab_r5f.txt; tail -2 $OUT/ab_r5f.txt
TORTOISE_MI355X_LIB=$WT timeout 400 python -m pytest tests/test_gpu_r4.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -k "two_engines or diffusion" > $OUT/r5f_wt_tests.log 2>&1; echo "wt tests rc=$?"; tail -3 $OUT/r5f_wt_tests.log
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?"; tail -4 $OUT/pytest_gpu.log
exit 0
