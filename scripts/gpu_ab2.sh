#!/bin/bash
# in-situ A/B of two builds on the denoiser stage: $1 = variant library name, repeated base / variant / base / variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
V=$PWD/tortoise_tts_amd/lib/libtortoise_mi355x_$1.so
STAGES=${2:-diff}
for r in 1 2; do
  AB_TAG=base timeout 300 python scripts/ab_stage.py $STAGES --reps 3 --dtype ${AB_DTYPE:-fp16} 2>&1 | grep "^ab "
  TORTOISE_MI355X_LIB=$V AB_TAG=$1 timeout 300 python scripts/ab_stage.py $STAGES --reps 3 --dtype ${AB_DTYPE:-fp16} 2>&1 | grep "^ab "
done
TORTOISE_MI355X_LIB=$V timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity_r3.py -q -m gpu -p no:cacheprovider -k "diffusion or schedules or sample_many" 2>&1 | tail -3
