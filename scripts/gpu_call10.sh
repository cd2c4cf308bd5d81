#!/bin/bash
# kernel microbenchmarks (experiments library): fabric / L2 bandwidth probe, GEMM tile variants at the denoiser and decode shapes, decode attention, flash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 900 python scripts/kbench.py ${KB:-bw gemm2 gemm_decode attn flash} 2>&1 | grep -v amdgpu > $OUT/kbench.txt
tail -5 $OUT/kbench.txt
