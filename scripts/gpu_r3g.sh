#!/bin/bash
# kernel trace of the 15-utterance decode batch (which kernels make the 14.7 ms step)
export TMPDIR=/tmp
mkdir -p gpurun_out/trace_g15
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/trace_g15 -o g15 --output-format csv -- python $GRAFT_REPO_ROOT/scripts/ab_groups.py 15 > $GRAFT_REPO_ROOT/gpurun_out/trace_g15/run.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/trace_g15 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/trace_g15_kernel_stats.csv
find gpurun_out/trace_g15 -name "*kernel_trace.csv" -delete
grep groups gpurun_out/trace_g15/run.log
head -12 gpurun_out/trace_g15_kernel_stats.csv | cut -c1-160
