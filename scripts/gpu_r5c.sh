#!/bin/bash
# round 5, call C: where does the denoiser's attention launch spend its time?  SQ counters over the microbenchmark (B = 2, H = 16, n = 870, relative
# positions) for the 32-query-wave kernel and the 16-query-wave kernel, plus the timing series of both.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
python -m tortoise_tts_amd.build --kbench > $OUT/kb_build.log 2>&1 || tail -5 $OUT/kb_build.log
for v in 1 0; do
  TT_FLASH_VARIANT=$v timeout 300 python scripts/kbench.py flash > $OUT/kb_flash_v$v.txt 2>&1
  cat $OUT/kb_flash_v$v.txt
  KB_FLASH_SHAPES=denoiser TT_FLASH_VARIANT=$v bash scripts/pmc.sh "flash" > $OUT/pmc_flash_v$v.log 2>&1
  cp $OUT/pmc/summary.txt $OUT/pmc_flash_v$v.txt
  cat $OUT/pmc_flash_v$v.txt
done
exit 0
