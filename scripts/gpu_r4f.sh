#!/bin/bash
# round 4: the complete bench.py --gpus 8 flow (32 candidates per rank, one all_gather, pair group inside the 8-rank world, ranks 2 - 7 skipping the
# tail, roofline leg on every rank) with the eight ranks SHARING the box's one GPU over gloo (TT_DIST_SHARE_DEVICE=1: control flow, not transport / timing)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 TT_DIST_SHARE_DEVICE=1 OMP_NUM_THREADS=8
OUT=gpurun_out
for n in 8 2; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_${n}rank_shared.log 2> $OUT/bench_${n}rank_shared.err
  echo "bench --gpus $n (shared device) rc=$?"
  tail -1 $OUT/bench_${n}rank_shared.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','n_gpus','ms_per_step','dtype')}, d['config']['parallelism'][:100])" || tail -5 $OUT/bench_${n}rank_shared.err
done
exit 0
