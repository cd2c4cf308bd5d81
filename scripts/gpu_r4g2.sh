#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
L=$PWD/tortoise_tts_amd/lib
for r in 1 2 3; do
  TT_DIFF_FUSED_GN=0 AB_TAG=apply timeout 300 python scripts/ab_stage.py diff --reps 3 --dtype fp16 2>&1 | grep "^ab "
  TT_DIFF_FUSED_GN=1 AB_TAG=fused timeout 300 python scripts/ab_stage.py diff --reps 3 --dtype fp16 2>&1 | grep "^ab "
  for v in "$@"; do
    TORTOISE_MI355X_LIB=$L/libtortoise_mi355x_$v.so TT_DIFF_FUSED_GN=1 AB_TAG=$v timeout 300 python scripts/ab_stage.py diff --reps 3 --dtype fp16 2>&1 | grep "^ab "
  done
done
