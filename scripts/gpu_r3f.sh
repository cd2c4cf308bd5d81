#!/bin/bash
# in-situ A/B of the tile thresholds on the batched decode (15 utterances x 256 candidates in one decode batch)
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/ab_r3f.txt
: > $out
L=tortoise_tts_amd/lib
for v in base t256 t128 tboth base; do
  if [ $v = base ]; then lib=$L/libtortoise_mi355x.so; else lib=$L/libtortoise_mi355x_$v.so; fi
  AB_NAME=$v TORTOISE_MI355X_LIB=$PWD/$lib timeout 300 python scripts/ab_groups.py 15 2>&1 | grep -E "groups|Error|error" >> $out
done
cat $out
