#!/bin/bash
# timing-only experiment: upper bound of what a pre-finalised GroupNorm statistics table would save (the variant skips the per-block
# reduction of the partial sums and computes WRONG values)
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/ab_r3j.txt
: > $out
L=$PWD/tortoise_tts_amd/lib
for v in base gnfake base gnfake; do
  if [ $v = base ]; then lib=$L/libtortoise_mi355x.so; else lib=$L/libtortoise_mi355x_$v.so; fi
  AB_TAG=$v TORTOISE_MI355X_LIB=$lib timeout 300 python scripts/ab_stage.py diff 2>&1 | grep -E "^ab |Error|error" >> $out
done
cat $out
