#!/bin/bash
# the complete --gpus 2 flow of bench.py at HEAD on ONE device (TT_DIST_SHARE_DEVICE=1: ranks share the GPU and exchange through gloo/host):
# a flow check of the N > 1 code path (sharded candidates, gather, split tail, chunk -> rank schedule), not a performance number
export TMPDIR=/tmp
O=gpurun_out
TT_DIST_SHARE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --no-roofline 2>$O/bench_2rank.err | tail -1 > $O/bench_2rank_shared.json
echo "utterance rc=$?"; cut -c1-400 $O/bench_2rank_shared.json; tail -2 $O/bench_2rank.err
TT_DIST_SHARE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --workload read --utterance-batch 4 --steps 1 --warmup 0 2>$O/bench_2rank_read.err | tail -1 > $O/bench_2rank_read_shared.json
echo "read rc=$?"; cut -c1-400 $O/bench_2rank_read_shared.json; tail -2 $O/bench_2rank_read.err
