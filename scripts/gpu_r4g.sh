#!/bin/bash
# round 4, g-3: ResBlock in_layers fused (GroupNorm + SiLU on the 1x1 GEMM's A path) - in-situ A/B of the denoiser stage, knob builds
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
L=$PWD/tortoise_tts_amd/lib
TT_DIFF_FUSED_GN=0 AB_TAG=apply timeout 300 python scripts/ab_stage.py diff --reps 3 --dtype fp16 2>&1 | grep "^ab "
TT_DIFF_FUSED_GN=1 AB_TAG=fused timeout 300 python scripts/ab_stage.py diff --reps 3 --dtype fp16 2>&1 | grep "^ab "
for v in "$@"; do
  TORTOISE_MI355X_LIB=$L/libtortoise_mi355x_$v.so TT_DIFF_FUSED_GN=1 AB_TAG=$v timeout 300 python scripts/ab_stage.py diff --reps 3 --dtype fp16 2>&1 | grep "^ab "
done
TT_DIFF_FUSED_GN=0 AB_TAG=apply timeout 300 python scripts/ab_stage.py diff --reps 3 --dtype fp16 2>&1 | grep "^ab "
TT_DIFF_FUSED_GN=1 AB_TAG=fused timeout 300 python scripts/ab_stage.py diff --reps 3 --dtype fp16 2>&1 | grep "^ab "
