#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?" | tee -a $OUT/summary.txt
grep -h "\[parity\]" $OUT/pytest_gpu.log > $OUT/parity.txt
timeout 600 python scripts/kbench.py attn gemm_decode > $OUT/kbench4.log 2>&1; echo "kbench rc=$?" | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench4.log 2> $OUT/bench4.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head; grep -v amdgpu $OUT/kbench4.log | grep "PRODUCT\|decode_attn" | grep -v "M= 32"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench4.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','stages_s_per_step')})
for r in d['kernel_breakdown_ms'][:14]: print(r)
PY
