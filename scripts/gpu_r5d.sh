#!/bin/bash
# round 5, call D: distribution-error / CLVP-ranking parity lines, flash32 with the clamp-free relative-position window (tests + in-situ A/B)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_r5.py -q -m gpu -s -p no:cacheprovider -k "distribution or ranking" > $OUT/r5d_parity.log 2>&1; echo "parity tests rc=$?"
grep -E "passed|failed|Error|assert|FULL" $OUT/r5d_parity.log | tail -30
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -p no:cacheprovider -k flash > $OUT/r5d_flash.log 2>&1; echo "flash tests rc=$?"; tail -3 $OUT/r5d_flash.log
timeout 400 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --flash-variants "1;0;1;0" > $OUT/ab_r5d.txt 2>&1; echo "ab rc=$?"
grep "^ab " $OUT/ab_r5d.txt; tail -2 $OUT/ab_r5d.txt
python -m tortoise_tts_amd.build --kbench > /dev/null 2>&1
for v in 1 0; do KB_FLASH_SHAPES=denoiser TT_FLASH_VARIANT=$v timeout 120 python scripts/kbench.py flash 2>&1 | grep flash; done
exit 0
