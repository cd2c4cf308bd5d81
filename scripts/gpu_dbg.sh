cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for n in 1 2 3 4 5 6 7; do TORTOISE_MI355X_LIB=$PWD/tortoise_tts_amd/lib/libtortoise_mi355x_ss$n.so AB_TAG=$n timeout 200 python scripts/dbg_sample_time.py 2>&1 | grep "^stop"; done
AB_TAG=full timeout 200 python scripts/dbg_sample_time.py 2>&1 | grep "^stop"
