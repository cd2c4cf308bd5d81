#!/bin/bash
# HBM traffic of the hot path's kernels from PMC counters, collected as MI355X_MICROARCH.md prescribes: one counter per
# pass (FETCH_SIZE and WRITE_SIZE do not fit one pass), --kernel-trace only, over one utterance of bench.py.
# hipGraph replay under counter collection segfaults in this rocprofv3, so the engines run eagerly (bench.py --no-graph): same kernels,
# same launch parameters.  Per-dispatch rows are aggregated ON the GPU box (the raw CSVs are too large to pull) into gpurun_out/pmc_bench.json.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT
# PMC_SETS: ';'-separated counter sets, one rocprofv3 pass each (default: the two HBM-traffic counters); PMC_JSON: output name
IFS=';' read -ra SETS <<< "${PMC_SETS:-FETCH_SIZE;WRITE_SIZE}"
i=0
for set in "${SETS[@]}"; do
  i=$((i+1)); c=pass$i
  (cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$c -o p -- \
     python $OLDPWD/bench.py --no-graph --steps 1 --warmup 0 --no-cpu-baseline --no-roofline ${PMC_BENCH_ARGS:---diffusion-iterations 24} > $OUT/$c.log 2>&1)
  echo "pmc [$set] rc=$?" | tee -a $OUT/summary.txt
done
OUT=$OUT PMC_JSON=${PMC_JSON:-pmc_bench.json} python - <<'PY'
import csv, glob, collections, json, os, re, sys
sys.path.insert(0, os.getcwd())
out = os.environ["OUT"]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))


from scripts.pmc_classes import klass


for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = klass(r["Kernel_Name"])
        if not name:
            continue
        a = agg[name][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
res = {}
for k, cs in agg.items():
    res[k] = {c: {"dispatches": v[0], "sum": v[1], "avg": v[1] / max(v[0], 1)} for c, v in cs.items()}
import bench
res["_meta"] = {"source_digest": bench.source_digest(), "command": "bench.py --no-graph --steps 1 --warmup 0 --no-cpu-baseline --no-roofline " + os.environ.get("PMC_BENCH_ARGS", "--diffusion-iterations 24") + " (every kernel class at its benchmark shape, 24 instead of 200 denoiser steps: this rocprofv3 segfaults in counter collection on the full 80 000-dispatch utterance)",
                "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch as rocprofv3 reports them (FETCH_SIZE is doubled by bench.py)"}
json.dump(res, open(out + "/../" + os.environ["PMC_JSON"], "w"), indent=1, sort_keys=True)
for k in sorted(res):
    if k != "_meta":
        print(k, {c: round(v["avg"], 1) for c, v in res[k].items()}, {c: v["dispatches"] for c, v in res[k].items()})
PY
find $OUT -name "*.csv" -size +1M -delete
