#!/bin/bash
# round 3, call A: new parity tests + the full -m gpu suite, load-path / ring-depth microbenchmarks, baseline bench line of this build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/summary.txt
rocminfo 2>/dev/null | grep -E "gfx|Marketing" | head -4 > $OUT/device.txt; nproc >> $OUT/device.txt
timeout 1500 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
grep -h "\[parity\]" $OUT/pytest_gpu.log | sed 's/^\.*//' > $OUT/parity.txt
grep -E "FULL-WIDTH|END-TO-END|S=2176|2186|cond_free=False|n=2176" $OUT/parity.txt
timeout 600 python scripts/kbench.py bw2 ring 2>&1 | grep -v amdgpu > $OUT/kbench_r3a.txt; echo "kbench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/kbench_r3a.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench_base.log 2> $OUT/bench_base.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
tail -1 $OUT/bench_base.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()})"
exit 0
