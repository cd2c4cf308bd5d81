#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 600 python scripts/kbench.py xcd > $OUT/kbench_xcd.log 2>&1; echo "kbench xcd rc=$?" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -k "gemm or full_diffusion or full_ar" > $OUT/pytest_gpu3.log 2>&1; echo "pytest-gpu3 rc=$?" | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench3.log 2> $OUT/bench3.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu3.log; grep -v amdgpu $OUT/kbench_xcd.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench3.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','stages_s_per_step')})
for r in d['kernel_breakdown_ms'][:12]: print(r)
PY
