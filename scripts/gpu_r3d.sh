export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stages.py tests/test_gpu_fullsize.py -q -m gpu -s -p no:cacheprovider -k "flash or diffusion or clvp or latents or prefill or cond" > gpurun_out/flash_tests.log 2>&1; echo rc=$?
grep -E "passed|failed|Error|assert" gpurun_out/flash_tests.log | tail -8; grep -E "parity.*flash" gpurun_out/flash_tests.log | head -12
for v in "" nosplit ""; do TORTOISE_MI355X_LIB=$PWD/tortoise_tts_amd/lib/libtortoise_mi355x${v:+_$v}.so timeout 300 python scripts/ab_stage.py diff --tag "${v:-split}" --reps 3 2>&1 | grep "^ab "; done | tee gpurun_out/ab_flash_split.txt
