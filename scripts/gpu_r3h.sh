#!/bin/bash
# kernel trace of the long-form workload (15 chunks, shared decode batch + shared padded denoiser passes)
export TMPDIR=/tmp
mkdir -p gpurun_out/trace_read
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/trace_read -o rd --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload read --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/trace_read/run.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/trace_read -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/trace_read_kernel_stats.csv
find gpurun_out/trace_read -name "*kernel_trace.csv" -delete
tail -1 gpurun_out/trace_read/run.log | cut -c1-300
head -16 gpurun_out/trace_read_kernel_stats.csv | cut -c1-170
