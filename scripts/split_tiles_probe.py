#!/usr/bin/env python
"""Review item 4 of round 5 ("denoiser GEMMs on 256-row / 128 x 128 tiles x split-K over the CUs"), measured without writing the fold:
the SLAB phase of a tile x split-K s form has exactly the workgroup count and per-workgroup feed / MFMA work of the same tile on
s * M rows at K / s (the K ranges become independent row blocks), which the experimental tiles of libtortoise_kbench.so can run as they are.
If that phase alone is not clearly below the product launch, the fold (>= one 4 - 5 us seam in this loop) cannot pay.
usage (GPU box): python scripts/split_tiles_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kbench as kb

kb.lib.tt_init()
M = 1740
SHAPES = [("qkv 1x1 1024->3072", 3072, 1024, 1, 0), ("proj / in_layers 1x1 1024->1024", 1024, 1024, 1, 0), ("k3 conv 1024->1024", 1024, 3072, 3, 870)]
TILES = [(600, "128x128 8w ring3", 128, 128), (1600, "128x128 8w ring2 2/CU", 128, 128), (2000, "256x128 8w ring3", 256, 128), (800, "256x64 8w ring3", 256, 64),
         (2200, "256x256 16w ring2", 256, 256)]
for name, N, K, taps, seq in SHAPES:
    us = kb.gemm_prod(M, N, K, taps, seq)
    print(f"split {name:32s} PRODUCT launch: {us:6.2f} us {kb.tf(M, N, K, us):6.1f} TFLOP/s", flush=True)
    for v, label, bm, bn in TILES:
        for s in (1, 2, 4):
            if (K // taps) % (64 * s):
                continue
            rows = M * s if not seq else seq * 2 * s  # conv: whole sequences per K range
            wgs = -(-rows // bm) * -(-N // bn)
            try:
                t = kb.gemm_exp(v, rows, N, K // s, taps, seq)
            except RuntimeError as ex:
                print(f"split {name:32s} {label:24s} split-K {s}: unsupported ({str(ex)[:60]})", flush=True)
                continue
            print(f"split {name:32s} {label:24s} split-K {s}: slab phase {t:6.2f} us  ({wgs:4d} workgroups)  vs product {us:6.2f}", flush=True)
