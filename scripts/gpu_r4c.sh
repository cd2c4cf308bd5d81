#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
L=tortoise_tts_amd/lib
for v in base attnpw nosc1; do
  if [ $v = base ]; then unset TORTOISE_MI355X_LIB; else export TORTOISE_MI355X_LIB=$PWD/$L/libtortoise_mi355x_$v.so; fi
  AB_TAG=$v timeout 300 python scripts/debug_ranges.py --rate 2>&1 | grep "^rate" | tee -a gpurun_out/debug_rate.txt
done
