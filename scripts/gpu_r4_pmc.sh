#!/bin/bash
# round 4: the two counter passes of scripts/gpu_r4_final.sh alone (classifier of scripts/pmc_bench.sh knows gemm_gna / EpiGeglu now; same build)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
bash scripts/pmc_bench.sh > $OUT/pmc_bench.log 2>&1; tail -3 $OUT/pmc_bench.log
PMC_SETS="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" PMC_JSON=pmc_sq.json bash scripts/pmc_bench.sh > $OUT/pmc_sq.log 2>&1; tail -2 $OUT/pmc_sq.log
exit 0
