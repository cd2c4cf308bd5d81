#!/bin/bash
# runtime environment knobs on the two graph-replayed stages (same library, same bits)
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/ab_r3k.txt
: > $out
run() { tag=$1; shift; env "$@" AB_TAG=$tag timeout 300 python scripts/ab_stage.py ar diff 2>&1 | grep -E "^ab |Error|error" >> $out; }
run base A=1
run devkernarg HIP_FORCE_DEV_KERNARG=1
run nointr HSA_ENABLE_INTERRUPT=0
run devk_off HIP_FORCE_DEV_KERNARG=0
run base A=1
cat $out
