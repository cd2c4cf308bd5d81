#!/bin/bash
# The ONE script that runs on the MI355X box (via gpurun).  usage: scripts/gpu.sh <task> [<task> ...]; every task has its own timeout and logs
# under gpurun_out/ so that a cut-off call still leaves evidence.  Tasks (parameters through the environment):
#   tests      pytest -m gpu                                   PYTEST_ARGS (default: tests), parity lines -> gpurun_out/parity.txt
#   smoke      __graft_entry__.smoke()
#   bench      python bench.py $BENCH_ARGS                      -> gpurun_out/bench.json (last line)
#   prof       rocprofv3 --kernel-trace --stats over bench.py   PROF_ARGS (default: --steps 1 --warmup 1 --no-cpu-baseline --no-roofline)
#   pmc        HBM traffic counters (FETCH_SIZE / WRITE_SIZE, one pass each) over one eager utterance -> gpurun_out/pmc_bench.json
#   pmc_sq     SQ counters over the same                         -> gpurun_out/pmc_sq.json
#   configs    the other BASELINE configurations                 -> gpurun_out/bench_other_configs.jsonl
#   workloads  --workload read / stream                          -> gpurun_out/bench_read.json, bench_stream.json
#   ranks      bench.py --gpus 2 / 4 / 8 UN-LAUNCHED (it starts its ranks itself) with the ranks sharing this box's one GPU over gloo
#   ab         python scripts/ab_stage.py $AB_ARGS               -> gpurun_out/ab.txt (appended)
#   kbench     python scripts/kbench.py $KBENCH_ARGS             -> gpurun_out/kbench.txt (appended)
#   py         python $PY_ARGS                                   -> gpurun_out/py.txt (appended)
#   final      pmc pmc_sq prof bench configs workloads tests smoke ranks
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
T0=$(date +%s)
rocminfo 2>/dev/null | grep -E "gfx|Marketing" | head -4 > $OUT/device.txt
nproc >> $OUT/device.txt
stamp() { echo "[gpu.sh] $1 done at $(( $(date +%s) - T0 )) s"; }

run_task() {
  case $1 in
    tests)
      timeout ${TESTS_TIMEOUT:-1500} python -m pytest ${PYTEST_ARGS:-tests} -q -m gpu -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest-gpu rc=$?"
      tail -3 $OUT/pytest_gpu.log; grep -E "^FAILED|^ERROR|Error|assert " $OUT/pytest_gpu.log | head -30
      grep "\[parity\]\|\[guard\]" $OUT/pytest_gpu.log | sed "s/^\.*//" > $OUT/parity.txt; wc -l $OUT/parity.txt ;;
    smoke)
      timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log ;;
    bench)
      timeout ${BENCH_TIMEOUT:-900} python bench.py ${BENCH_ARGS:-} > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"
      tail -1 $OUT/bench.log > $OUT/bench.json
      python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench.json"))
    print({k: d[k] for k in ("metric", "value", "ms_per_step", "dtype", "n_gpus")}, {k: round(v, 4) for k, v in d["stages_s_per_step"].items()})
    r = d.get("roofline")
    if r:
        print({k: r.get(k) for k in ("kernel", "frac", "achieved", "traffic", "traffic_stale", "avg_launch_us", "algorithmic_bytes_per_launch")})
        print({k: round(v["frac"], 4) for k, v in r.get("stages", {}).items() if v.get("frac")})
    c = d.get("cpu_baseline")
    if c:
        print("cpu", c["value"], c["cores"], (c.get("reference_ratio") or {}).get("utterance"))
except Exception as e:
    print("bench line unreadable:", e); print(open("gpurun_out/bench.err").read()[-2000:])
PY
      ;;
    prof)
      rm -rf $OUT/prof
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o prof -- python $OLDPWD/bench.py ${PROF_ARGS:---steps 1 --warmup 1 --no-cpu-baseline --no-roofline} > $OLDPWD/$OUT/prof.log 2>&1); echo "prof rc=$?"
      find $OUT/prof -type f -size +8M -delete; find $OUT/prof -name "*kernel_stats.csv" | head -3; tail -2 $OUT/prof.log ;;
    pmc)
      timeout 700 bash scripts/pmc_bench.sh > $OUT/pmc_bench.log 2>&1; tail -4 $OUT/pmc_bench.log
      # the bench line of THIS call reads the traffic of THIS build (the committed copy is made from gpurun_out/ afterwards)
      [ -f $OUT/pmc_bench.json ] && cp $OUT/pmc_bench.json profiles/${ROUND_TAG:-r06}_pmc_bench.json ;;
    pmc_sq)
      PMC_SETS="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" PMC_JSON=pmc_sq.json \
        timeout 450 bash scripts/pmc_bench.sh > $OUT/pmc_sq.log 2>&1; tail -3 $OUT/pmc_sq.log ;;
    configs)
      : > $OUT/bench_other_configs.jsonl
      for cfg in "--preset fast" "--preset high_quality" "--mel-tokens 500" "--dtype fp16" "--dtype bf16" "--preset ultra_fast" "--candidates-per-rank 32" "--candidates-per-rank 64" "--candidates-per-rank 128"; do
        timeout 240 python bench.py $cfg --steps 3 --warmup 1 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/bench_other_configs.jsonl
      done
      python - <<'PY'
import json
for l in open("gpurun_out/bench_other_configs.jsonl"):
    try:
        d = json.loads(l); print(d["metric"], d["config"]["workload"][:52], d["dtype"], round(d["value"], 2), round(d["latency_s"], 4), {k: round(v, 4) for k, v in d["stages_s_per_step"].items()})
    except Exception as e:
        print("unreadable line:", e)
PY
      ;;
    workloads)
      timeout 300 python bench.py --workload read --steps 1 --warmup 1 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_read.json
      timeout 200 python bench.py --workload stream --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_stream.json
      python -c "
import json
for f in ('bench_read', 'bench_stream'):
    d = json.load(open('$OUT/' + f + '.json')); print(f, round(d['value'], 2), round(d['ms_per_step'], 1), d.get('first_chunk_latency_s'))" ;;
    ranks)
      for n in ${RANKS:-2 4 8}; do
        TT_DIST_SHARE_DEVICE=1 OMP_NUM_THREADS=8 timeout 480 python bench.py --gpus $n --steps 1 --warmup 1 --no-cpu-baseline ${RANKS_ARGS:-} > $OUT/bench_${n}rank_shared.log 2> $OUT/bench_${n}rank_shared.err
        echo "python bench.py --gpus $n (un-launched, shared device) rc=$?"
        tail -1 $OUT/bench_${n}rank_shared.log | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value', 'n_gpus', 'ms_per_step', 'replica_rtf', 'rccl_ranks_seen', 'collective_backend', 'all_gather_calls_per_utterance')}, d['config']['parallelism'][:100])" || tail -5 $OUT/bench_${n}rank_shared.err
      done ;;
    ab)
      timeout ${AB_TIMEOUT:-900} python scripts/ab_stage.py ${AB_ARGS:-ar} 2>&1 | grep -v "^\[" | tee -a $OUT/ab.txt | tail -${AB_TAIL:-40} ;;
    kbench)
      timeout ${KBENCH_TIMEOUT:-900} python scripts/kbench.py ${KBENCH_ARGS:-} 2>&1 | tee -a $OUT/kbench.txt | tail -${KBENCH_TAIL:-60} ;;
    py)
      timeout ${PY_TIMEOUT:-900} python ${PY_ARGS} 2>&1 | tee -a $OUT/py.txt | tail -${PY_TAIL:-60} ;;
    final)
      for t in pmc pmc_sq prof bench configs workloads tests smoke ranks; do run_task $t; stamp $t; done ;;
    *) echo "unknown task $1" ;;
  esac
}

for task in "$@"; do run_task $task; stamp $task; done
exit 0
