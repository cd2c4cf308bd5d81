#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
L=$PWD/tortoise_tts_amd/lib
for v in "$@"; do
TORTOISE_MI355X_LIB=$L/libtortoise_mi355x_$v.so timeout 600 python -m pytest tests/test_gpu_r4.py -q -m gpu -x -s -k "fused_groupnorm" -p no:cacheprovider 2>&1 | grep -E "parity|passed|failed|Error|error|assert" | head -20
done
bash scripts/gpu_r4g2.sh "$@"
