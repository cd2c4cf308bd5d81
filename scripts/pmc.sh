#!/bin/bash
# PMC passes over the kernel microbenchmarks (counters in their own runs, kernel-trace only).  usage: scripts/pmc.sh "<kbench args>"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
ARGS="${1:-flash tiles}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $OLDPWD/scripts/kbench.py $ARGS > $OUT/p$i.log 2>&1)
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "gpurun_out/pmc")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:110]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as o:
    for k, cs in sorted(agg.items()):
        if not ("gemm" in k or "flash" in k or "decode_attn" in k or "gn_" in k): continue
        n = max(len(v) for v in cs.values())
        o.write(f"{k}  (dispatches {n})\n")
        for c, v in sorted(cs.items()):
            o.write(f"    {c:28s} avg {sum(v)/len(v):16.1f}\n")
print(open(out + "/summary.txt").read()[:6000])
PY
find $OUT -name "*.csv" -size +4M -delete
