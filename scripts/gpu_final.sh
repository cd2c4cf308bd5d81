#!/bin/bash
# measurement pass of a build: full gpu test suite, driver-style bench line, kernel trace, HBM-traffic PMC passes, microbenchmarks
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/summary.txt
timeout 1200 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/pytest_gpu.log; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head
grep -h "\[parity\]" $OUT/pytest_gpu.log | sed 's/^\.*//' > $OUT/parity.txt
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()})
print(d['roofline']); print({k:v for k,v in d['cpu_baseline'].items() if k!='sample'})
for r in d['kernel_breakdown_ms'][:16]: print('   %-40s %6d %9.3f %8.2f' % (r['kernel'], r['launches'], r['total_ms'], r['avg_us']))
PY
bash scripts/gpu_round.sh prof > $OUT/prof_phase.log 2>&1; tail -3 $OUT/prof_phase.log
bash scripts/pmc_bench.sh > $OUT/pmc_phase.log 2>&1; tail -22 $OUT/pmc_phase.log
timeout 600 python scripts/kbench.py bw gemm2 gemm_decode attn flash 2>&1 | grep -v amdgpu > $OUT/kbench.txt; tail -3 $OUT/kbench.txt
for cfg in "--preset fast" "--preset high_quality --dtype fp16" "--mel-tokens 500" "--dtype fp16"; do
  timeout 600 python bench.py $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> $OUT/bench_other.jsonl
done
timeout 600 python bench.py --workload read --steps 1 --warmup 0 2>/dev/null | tail -1 > $OUT/bench_read.json
TT_DIST_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --no-roofline 2>$OUT/bench_2rank.err | tail -1 > $OUT/bench_2rank_shared.json
python - <<'PY'
import json
for f in ('gpurun_out/bench_read.json', 'gpurun_out/bench_2rank_shared.json'):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['metric'], d['n_gpus'], round(d['ms_per_step'],1), 'ms RTF', round(d['value'],2))
    except Exception as e:
        print(f, 'FAILED', e)
for l in open('gpurun_out/bench_other.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:70], d['dtype'], round(d['ms_per_step'],1), 'ms RTF', round(d['value'],2))
PY
