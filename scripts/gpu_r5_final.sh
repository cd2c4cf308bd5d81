#!/bin/bash
# round 5, final measurement pass of ONE build: PMC traffic (FETCH / WRITE) + SQ counters over the benchmark's kernels, kernel trace, the default
# bench line (roofline + cpu_baseline; traffic from the PMC pass of this very call), the other BASELINE configurations, read / stream workloads,
# the -m gpu suite with its [parity] lines.  Every step has its own timeout.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
date +%s > $OUT/final_t0
timeout 700 bash scripts/pmc_bench.sh > $OUT/pmc_bench.log 2>&1; tail -4 $OUT/pmc_bench.log
[ -f $OUT/pmc_bench.json ] && cp $OUT/pmc_bench.json profiles/r05_pmc_bench.json
echo "after pmc traffic: $(( $(date +%s) - $(cat $OUT/final_t0) )) s"
PMC_SETS="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" PMC_JSON=pmc_sq.json timeout 450 bash scripts/pmc_bench.sh > $OUT/pmc_sq.log 2>&1; tail -3 $OUT/pmc_sq.log
echo "after pmc sq: $(( $(date +%s) - $(cat $OUT/final_t0) )) s"
PROF_ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-roofline" timeout 300 bash scripts/gpu_round.sh prof > $OUT/prof_round.log 2>&1; tail -2 $OUT/prof_round.log
echo "after trace: $(( $(date +%s) - $(cat $OUT/final_t0) )) s"
timeout 600 python bench.py > $OUT/bench_final.log 2> $OUT/bench_final.err; echo "bench rc=$?"
tail -1 $OUT/bench_final.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','dtype')}, {k:round(v,4) for k,v in d['stages_s_per_step'].items()}); r=d['roofline']; print({k:r[k] for k in ('kernel','frac','achieved','traffic','traffic_stale','avg_launch_us','algorithmic_bytes_per_launch')}); print({k:round(v['frac'],4) for k,v in r['stages'].items() if v['frac']}); c=d['cpu_baseline']; print(c['value'], c['cores'], c['reference_ratio'])"
echo "after bench: $(( $(date +%s) - $(cat $OUT/final_t0) )) s"
: > $OUT/bench_other_configs.jsonl
for cfg in "--preset fast" "--preset high_quality" "--mel-tokens 500" "--dtype fp16" "--dtype bf16" "--preset ultra_fast"; do
  timeout 200 python bench.py $cfg --steps 2 --warmup 1 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/bench_other_configs.jsonl
done
python -c "
import json
for l in open('$OUT/bench_other_configs.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:60], d['dtype'], round(d['value'],2), round(d['latency_s'],3))"
timeout 300 python bench.py --workload read --steps 1 --warmup 1 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_read_final.json
timeout 200 python bench.py --workload stream --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_stream_final.json
python -c "
import json
for f in ('bench_read_final','bench_stream_final'):
    d=json.load(open('$OUT/'+f+'.json')); print(f, round(d['value'],2), round(d['ms_per_step'],1), d.get('first_chunk_latency_s'))"
echo "after workloads: $(( $(date +%s) - $(cat $OUT/final_t0) )) s"
timeout 600 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $OUT/pytest_gpu_final.log 2>&1; echo "pytest-gpu rc=$?"; tail -2 $OUT/pytest_gpu_final.log
grep "\[parity\]\|\[guard\]" $OUT/pytest_gpu_final.log | sed "s/^\.*//" > $OUT/parity_final.txt; wc -l $OUT/parity_final.txt
timeout 120 python __graft_entry__.py smoke > $OUT/smoke_final.log 2>&1; tail -2 $OUT/smoke_final.log
# the complete multi-rank flows with the ranks sharing the box's one GPU over gloo (control flow only; the numbers mean nothing)
for n in 2 8; do
  TT_DIST_SHARE_DEVICE=1 OMP_NUM_THREADS=8 timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2962$n bench.py --gpus $n --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_${n}rank_shared.log 2> $OUT/bench_${n}rank_shared.err
  echo "bench --gpus $n (shared device) rc=$?"
  tail -1 $OUT/bench_${n}rank_shared.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','n_gpus','ms_per_step','dtype')}, d['config']['parallelism'][:100])" || tail -5 $OUT/bench_${n}rank_shared.err
done
echo "total: $(( $(date +%s) - $(cat $OUT/final_t0) )) s"
exit 0
