#!/bin/bash
# validation + measurement pass of a changed GEMM build within a small GPU budget: the whole gpu suite first (stop on failure),
# then the HBM-traffic PMC passes of THIS build, then the default bench line against them
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 150 python -m pytest tests -q -m gpu -s -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; rc=$?
echo "pytest-gpu rc=$rc"; tail -1 $OUT/pytest_gpu.log; grep -E "^FAILED|^ERROR|Error" $OUT/pytest_gpu.log | head -5
grep -h "\[parity\]" $OUT/pytest_gpu.log | sed 's/^\.*//' > $OUT/parity.txt
[ $rc -ne 0 ] && exit 1
bash scripts/pmc_bench.sh > $OUT/pmc_phase.log 2>&1; tail -4 $OUT/pmc_phase.log
cp $OUT/pmc_bench.json profiles/r02_pmc_bench.json
timeout 120 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()})
print({k:d['roofline'][k] for k in ('kernel','frac','traffic','traffic_stale')})
for r in d['kernel_breakdown_ms'][:18]: print('   %-40s %6d %9.3f %8.2f' % (r['kernel'], r['launches'], r['total_ms'], r['avg_us']))
PY
