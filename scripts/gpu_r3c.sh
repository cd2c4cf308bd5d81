#!/bin/bash
# round 3, call C: full -m gpu suite, headline bench, AR stage alone, read bench at 16 chunks per batch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
grep -h "\[parity\]" $OUT/pytest_gpu.log | sed 's/^\.*//' > $OUT/parity.txt
grep -E "FULL-WIDTH [0-9]+-iteration" $OUT/parity.txt
timeout 300 python scripts/ab_stage.py ar diff --reps 3 2>&1 | grep "^ab " | tee $OUT/ab_stage.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
tail -1 $OUT/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()}); print([(k['kernel'],k['avg_us']) for k in d['kernel_breakdown_ms'][:14]])"
timeout 500 python bench.py --workload read --steps 1 --warmup 1 --utterance-batch 16 --no-roofline --no-cpu-baseline 2>$OUT/bench_read_ub16.err | tail -1 > $OUT/bench_read_ub16.json
python -c "
import json
d=json.load(open('$OUT/bench_read_ub16.json')); print('read ub16', d['value'], d['ms_per_step'], d['stages_s_per_step'])"
exit 0
