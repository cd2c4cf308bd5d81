#!/bin/bash
# round 5, call H: the AttentionBlock norm on the QKV GEMM's A path (TT_DIFF_OPT_FUSED_GN = 2) and out_layers' norm in the in_layers launch (3) - parity test, in-situ A/B on one diffusion stage
# object (values alternating), and a kernel trace of the fused form for the per-launch durations
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_r5.py tests/test_gpu_r4.py -q -m gpu -s -p no:cacheprovider -k "more_groupnorm_fusions or fused_groupnorm" > $OUT/r5h_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error|assert|\[parity\]" $OUT/r5h_tests.log | tail -8
timeout 400 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --gn-variants "1;3;1;3;1;3;4;2;0" > $OUT/ab_r5h_gn.txt 2>&1; echo "ab gn rc=$?"; grep "^ab " $OUT/ab_r5h_gn.txt
TT_DIFF_OVERLAP_PREPASS=0 timeout 300 python scripts/ab_stage.py diff --dtype fp16 --reps 2 --gn-variants "1;3;1;3" > $OUT/ab_r5h_gn_serial.txt 2>&1; echo "ab gn (pre-pass first) rc=$?"; grep "^ab " $OUT/ab_r5h_gn_serial.txt
rm -rf $OUT/prof_r5h
TT_DIFF_FUSED_GN=3 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r5h -o gn3 --output-format csv -- python scripts/ab_stage.py diff --dtype fp16 --reps 1 --iterations 100 > $OUT/prof_r5h.log 2>&1; echo "trace rc=$?"
f=$(find $OUT/prof_r5h -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { cp "$f" $OUT/r5h_kernel_stats_gn3.csv; head -12 "$f" | cut -c1-200; }
find $OUT/prof_r5h -name "*kernel_trace.csv" -delete
exit 0
