#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_full.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py "tests/test_gpu_stages.py" -q -m gpu -s -p no:cacheprovider -k "cached_decode or full_ar or sample or generate or sampling or gemm" > $OUT/pytest_gpu2.log 2>&1; echo "pytest-gpu2 rc=$?" | tee -a $OUT/summary.txt
timeout 600 python scripts/kbench.py bw > $OUT/kbench_bw.log 2>&1; echo "kbench bw rc=$?" | tee -a $OUT/summary.txt
timeout 600 python scripts/kbench.py attn > $OUT/kbench_attn.log 2>&1; echo "kbench attn rc=$?" | tee -a $OUT/summary.txt
timeout 600 python scripts/kbench.py gemm2 gemm_decode > $OUT/kbench_gemm2.log 2>&1; echo "kbench gemm2 rc=$?" | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench2.log 2> $OUT/bench2.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu2.log; grep -E "FAILED|Error" $OUT/pytest_gpu2.log | head -20; tail -c 1200 $OUT/bench2.log
