#!/bin/bash
# round 3, call B: in-situ A/B of the decode-step knobs (csrc/knobs.h) on the AR stage at the benchmark shape; every build in ONE call
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
L=tortoise_tts_amd/lib
: > $OUT/ab_r3b.txt
for v in "" ${AB_VARIANTS:-nt rid rid128 sc1 kvnt all} ""; do
  lib=$L/libtortoise_mi355x${v:+_$v}.so
  TORTOISE_MI355X_LIB=$PWD/$lib timeout 300 python scripts/ab_stage.py ${AB_STAGES:-ar} --tag "${v:-base}" --reps ${AB_REPS:-3} 2>&1 | grep -E "^ab |Error|error" | tee -a $OUT/ab_r3b.txt
done
if [ "${AB_PYTEST:-1}" = "1" ]; then
  timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest_gpu_b.log 2>&1; echo "pytest-gpu rc=$?" | tee -a $OUT/ab_r3b.txt
  tail -3 $OUT/pytest_gpu_b.log
fi
exit 0
