#!/bin/bash
# flash attention: microbenchmark + SQ / LDS counters over it (counters in their own passes, kernel-trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_flash
rm -rf $OUT; mkdir -p $OUT
timeout 200 python scripts/kbench.py flash 2>&1 | grep -v amdgpu | tee $OUT/kbench_flash.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $OLDPWD/scripts/kbench.py flash > $OUT/p$i.log 2>&1)
  echo "pass $i rc=$?"
done
OUT=$OUT python - <<'PY'
import csv, glob, collections, os
out = os.environ["OUT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70] + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as o:
    for k, cs in sorted(agg.items()):
        if "flash" not in k: continue
        n = max(len(v) for v in cs.values())
        o.write(f"{k}  (dispatches {n})\n")
        for c, v in sorted(cs.items()):
            o.write(f"    {c:28s} avg {sum(v)/len(v):16.1f}\n")
print(open(out + "/summary.txt").read()[:7000])
PY
find $OUT -name "*.csv" -size +1M -delete
