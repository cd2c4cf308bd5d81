#!/bin/bash
# HBM traffic of the hot path's kernels from PMC counters, collected as MI355X_MICROARCH.md prescribes: one counter per
# pass (FETCH_SIZE and WRITE_SIZE do not fit one pass), --kernel-trace only, over one utterance of bench.py.
# hipGraph replay under counter collection segfaults in this rocprofv3, so the engines run eagerly (TT_NO_GRAPH=1): same kernels,
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
     env TT_NO_GRAPH=1 python $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/$c.log 2>&1)
  echo "pmc [$set] rc=$?" | tee -a $OUT/summary.txt
done
OUT=$OUT PMC_JSON=${PMC_JSON:-pmc_bench.json} python - <<'PY'
import csv, glob, collections, json, os, re
out = os.environ["OUT"]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"(gemm_glds_kernel|gemm_kernel|flash_kernel|decode_attn_kernel|gn_apply\w*|gn_stats\w*|rownorm\w*|sample_kernel|lvc_kernel)", k)
        if not m:
            continue
        name = m.group(1)
        if name.startswith("gemm"):
            t = re.search(r"Li(\d+)ELi(\d+)E", k)
            e = re.search(r"(EpiStd|EpiQkvHeads|EpiQkvDecode)", k)
            name = "gemm<%s,%s,%s>" % (t.group(1) if t else "?", t.group(2) if t else "?", e.group(1) if e else "?")
        a = agg[name][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
res = {}
for k, cs in agg.items():
    res[k] = {c: {"dispatches": v[0], "sum": v[1], "avg": v[1] / max(v[0], 1)} for c, v in cs.items()}
json.dump(res, open(out + "/../" + os.environ["PMC_JSON"], "w"), indent=1, sort_keys=True)
for k in sorted(res):
    print(k, {c: round(v["avg"], 1) for c, v in res[k].items()}, {c: v["dispatches"] for c, v in res[k].items()})
PY
find $OUT -name "*.csv" -size +1M -delete
