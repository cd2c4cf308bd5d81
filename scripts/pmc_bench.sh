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


def klass(k):
    """rocprofv3 kernel name -> the engine profiler's class name (tortoise_tts_amd/csrc/common.hip g_prof_names)."""
    if "gemm_gna_kernel" in k:
        return "gemm_gna<32,256,EpiStd,stats>"
    m = re.search(r"gemm_glds_kernelI\w+?Li(\d+)ELi(\d+)ELi\d+ELi\d+ELi\d+ENS_\d+(EpiStd|EpiQkvHeads|EpiQkvDecode|EpiGeglu)(\w*?)EELb([01])ELb[01]E", k)
    if m:
        bm, bn, epi, targs, conv = m.groups()
        # EpiStd<T, ACT, STATS, MODE>: the 64x64 1x1 GEMMs with the statistics epilogue (denoiser) are their own class in the engine's profiler
        st = re.match(r"IDF16[b_]Lin?\d+ELi1E", targs) is not None
        if epi == "EpiStd" and bm == "64" and bn == "64" and conv == "0" and st:
            return "gemm_glds<64,64,EpiStd,1x1,stats>"
        if epi == "EpiGeglu":
            return "gemm_glds<%s,%s,EpiStd,1x1>" % (bm, bn)  # (reported with the plain 1x1 class of its tile, as the engine's profiler does)
        return "gemm_glds<%s,%s,%s%s>" % (bm, bn, epi, (",conv" if conv == "1" else ",1x1") if epi == "EpiStd" else "")
    if "gemm_conv3s_kernel" in k:
        return "gemm_glds<128,64,EpiStd,conv>"  # the shared-halo 3-tap kernel reports under the conv class of its tile
    if "gemm_glds_kernel" in k:
        m = re.search(r"gemm_glds_kernel<[^,]+, (\d+), (\d+), \d+, \d+, \d+, tt::(\w+)<[^>]+>, (true|false)", k)
        if m:
            bm, bn, epi, conv = m.groups()
            return "gemm_glds<%s,%s,%s%s>" % (bm, bn, epi, (",conv" if conv == "true" else ",1x1") if epi == "EpiStd" else "")
        return "gemm_glds<?>"
    for pat, name in (("flash_lds_kernel", "flash_kernel"), ("flash_kernel", "flash_kernel"), ("decode_attn_lds_kernel", "decode_attn_kernel"), ("decode_attn_kernel", "decode_attn_kernel"),
                      ("gn_apply", "gn_apply_kernel(+gn_stats)"), ("gn_stats", "gn_apply_kernel(+gn_stats)"), ("rownorm", "rownorm_kernel"),
                      ("sample_kernel", "sample_kernel"), ("lvc_kernel", "lvc_kernel"), ("conv1d_direct", "conv1d_direct_kernel"), ("convt1d", "convt1d_kernel")):
        if pat in k:
            return name
    return None


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
