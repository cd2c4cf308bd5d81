#!/bin/bash
export TMPDIR=/tmp
L=$PWD/tortoise_tts_amd/lib
AB_TAG=gnfake TORTOISE_MI355X_LIB=$L/libtortoise_mi355x_gnfake.so timeout 300 python scripts/ab_stage.py diff 2>&1 | tail -15
