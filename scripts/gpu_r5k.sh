#!/bin/bash
# round 5, call K: per-kernel durations of the 256 x 256 tile users with the eight-phase / the 16-wave kernel on ONE box: kernel traces of the default
# bench line (pre-pass serialised in front of the sampler loop so that its kernels run alone), TT_GEMM_VARIANT = 1 / 0
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
for v in 1 0; do
  rm -rf $OUT/prof_r5k_$v
  TT_GEMM_VARIANT=$v TT_DIFF_OVERLAP_PREPASS=0 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_r5k_$v -o t --output-format csv -- python bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline > $OUT/prof_r5k_$v.log 2>&1; echo "trace variant $v rc=$?"
  f=$(find $OUT/prof_r5k_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/r5k_kernel_stats_variant$v.csv
  find $OUT/prof_r5k_$v -name "*kernel_trace.csv" -delete
  tail -1 $OUT/prof_r5k_$v.log | cut -c1-200
done
python - <<'PY'
import csv
for v in (1, 0):
    rows = list(csv.DictReader(open('gpurun_out/r5k_kernel_stats_variant%d.csv' % v)))
    print('--- variant', v)
    for r in rows:
        n = r['Name']
        if 'gemm_p8' in n or 'Li256ELi256' in n or ('256, 256' in n):
            print('%-110s %5s %10.1f us' % (n[:110], r['Calls'], float(r['AverageNs']) / 1e3))
PY
exit 0
