#!/bin/bash
# round-2 baseline of this session: full gpu test suite, driver-style bench line, kernel trace, HBM-traffic PMC passes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/summary.txt
timeout 1200 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head
grep -h "\[parity\]" $OUT/pytest_gpu.log | tail -200 > $OUT/parity.txt
timeout 600 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()})
print(d['roofline']); print(d['cpu_baseline'])
for r in d['kernel_breakdown_ms'][:25]: print('   %-40s %6d %9.3f %8.2f' % (r['kernel'], r['launches'], r['total_ms'], r['avg_us']))
PY
bash scripts/gpu_round.sh prof > $OUT/prof_phase.log 2>&1; tail -5 $OUT/prof_phase.log
bash scripts/pmc_bench.sh > $OUT/pmc_phase.log 2>&1; tail -20 $OUT/pmc_phase.log
