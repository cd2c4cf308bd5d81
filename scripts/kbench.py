#!/usr/bin/env python
"""Kernel experiments on the MI355X (run through gpurun): drives tortoise_tts_amd/lib/libtortoise_kbench.so
(`python -m tortoise_tts_amd.build --kbench`; sources in tortoise_tts_amd/csrc/kbench/).  Every number is device time per
launch from a replayed hipGraph of a launch chain, on pseudo-random (full-sign) operands.
usage: scripts/kbench.py [gemm_denoiser] [gemm_decode] [attn] ..."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (binds the HIP runtime the library links against)

lib = C.CDLL(os.path.join(ROOT, "tortoise_tts_amd", "lib", "libtortoise_kbench.so"))
lib.tt_last_error.restype = C.c_char_p
D = C.c_double


def chk(rc):
    if rc != 0:
        raise RuntimeError(lib.tt_last_error().decode())


def gemm_exp(variant, M, N, K, taps=1, seq=0, nw=1, chain=32, pad=0, reps=10):
    us = D(0)
    chk(lib.tt_kb_gemm_exp(variant, M, N, K, taps, seq, nw, chain, pad, reps, C.byref(us)))
    return us.value


def gemm_prod(M, N, K, taps=1, seq=0, splitk=1, packed=0, nw=1, na=1, xcd_rows=0, chain=32, reps=10, act=0):
    us = D(0)
    chk(lib.tt_kb_gemm_prod(M, N, K, taps, seq, splitk, packed, nw, na, xcd_rows, chain, reps, C.byref(us), act))
    return us.value


def attn(variant, B, H, P1, tgen, tmax, nl, chain=16, reps=10):
    us, md = D(0), D(0)
    chk(lib.tt_kb_decode_attn(variant, B, H, P1, tgen, tmax, nl, chain, reps, C.byref(us), C.byref(md)))
    return us.value, md.value


def gemm_exp_na(variant, M, N, K, nw=1, na=1, chain=32, reps=10):
    us = D(0)
    chk(lib.tt_kb_gemm_exp_na(variant, M, N, K, nw, na, chain, reps, C.byref(us)))
    return us.value


CFG = {0: "128x64 8w(2x4) ring4 [product tile]", 1: "128x64 4w(2x2) ring4", 3: "128x64 8w(4x2) ring4", 4: "64x64 4w ring4 [product decode tile]",
       6: "128x128 8w(2x4) ring3", 7: "128x128 4w(2x2) ring3", 8: "256x64 8w(4x2) ring3", 9: "128x64 8w(2x4) ring6",
       30: "64x64 4w ring8", 31: "64x64 4w ring10", 32: "32x32 4w ring16", 33: "32x32 4w ring20", 34: "64x64 8w(4x2) ring10", 35: "64x64 4w ring6",
       36: "32x64 4w ring12"}
MODE = {0: "full", 1: "no loads in k-loop", 2: "no LDS reads / MFMA"}


def tf(M, N, K, us):
    return 2.0 * M * N * K / us / 1e6


def bw_probe(mode, waves, nblocks, footprint, bytes_per_wg, reps=10):
    us = D(0)
    lib.tt_kb_bw_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(D)]
    chk(lib.tt_kb_bw_probe(mode, waves, nblocks, footprint, bytes_per_wg, reps, C.byref(us)))
    return us.value


EXTRA = {200: "128x64 4w ring3 2 blocks/CU", 500: "64x64 4w ring3 2 blocks/CU", 501: "64x64 4w ring2 4 blocks/CU", 502: "64x64 4w ring3 3 blocks/CU",
         1300: "128x64 8w(2x4) ring3 2 blocks/CU", 1301: "128x64 8w(4x2) ring3 2 blocks/CU", 1600: "128x128 8w ring2 2 blocks/CU",
         2000: "256x128 8w(4x2) ring3", 2100: "256x128 16w(4x4) ring3", 2200: "256x256 16w(4x4) ring2", 2300: "256x256 8w(2x4) ring2",
         2400: "128x256 8w(2x4) ring3"}


def concurrency(M, chain=30, tgen=100, reps=20):
    out = (D * 8)()
    chk(lib.tt_kb_concurrency(M, chain, tgen, reps, out))
    return list(out)[:7]


def main():
    which = sys.argv[1:] or ["bw", "gemm2", "gemm_decode", "attn"]
    lib.tt_init()
    if "streams" in which:
        # two launch chains on two streams: do kernels of different queues share the chip?  (row ranges of the decode step)
        print("two-stream concurrency probe (us per replay of a 30-launch chain; G = decode projection GEMM N = K = 1024 split-K 4, A = decode attention 16 heads x 100 own keys)")
        for M in (64, 128, 256):
            g, gg, a, ag, aa, gg1, ag1 = concurrency(M)
            print("  M=%3d: G alone %7.1f | G||G two streams %7.1f (x%.2f of one; 1.0 = full overlap, 2.0 = serialised) | A alone %7.1f | A||G %7.1f "
                  "(sum %.1f, max %.1f) | A||A %7.1f (x%.2f) | one graph, two branches: G||G %7.1f, A||G %7.1f" %
                  (M, g, gg, gg / g, a, ag, a + g, max(a, g), aa, aa / a, gg1, ag1), flush=True)
    if "bw" in which:
        for fp, tag in ((2 << 20, "2 MiB (one L2)"), (24 << 20, "24 MiB (all L2s)"), (160 << 20, "160 MiB (Infinity Cache)"), (2 << 30, "2 GiB (HBM)")):
            for mode, mtag in ((0, "global_load_lds 16B"), (1, "global_load_dwordx4 -> VGPR, full lines"), (2, "global_load_dwordx4 -> VGPR, fragment shaped")):
                for waves in (4, 8):
                    for nb in (256, 512):
                        per = 1 << 20
                        us = bw_probe(mode, waves, nb, fp, per)
                        print(f"bw_probe {tag:24s} {mtag:44s} {waves} waves x {nb} blocks: {us:8.2f} us  {nb * per / us / 1e6:7.2f} TB/s  {per * (nb / 256) / us / 1e3:6.1f} GB/s per CU", flush=True)
    if "flash" in which:
        if os.environ.get("TT_FLASH_VARIANT"):  # 1 = 32-query waves (flash32_kernel), 0 = 16-query waves
            lib.ttx_kernel_variant(0, int(os.environ["TT_FLASH_VARIANT"]))  # TTX_FLASH32
        shapes = ((2, 16, 870, 0, 1), (2, 16, 2176, 0, 1), (32, 16, 870, 0, 1), (256, 12, 200, 0, 0), (1, 16, 260, 1, 0))
        if os.environ.get("KB_FLASH_SHAPES") == "denoiser":
            shapes = shapes[:1]
        for (B, H, n, causal, rel) in shapes:
            us = D(0)
            chk(lib.tt_kb_flash(B, H, n, causal, rel, 16, 10, C.byref(us)))
            fl = 4.0 * B * H * n * n * 64 * (0.5 if causal else 1.0)
            print(f"flash B={B} H={H} n={n} causal={causal} relpos={rel}: {us.value:8.2f} us {fl / us.value / 1e6:7.1f} TFLOP/s", flush=True)
    if "xcd" in which:
        shapes = [("1x1 1024->1024", 1740, 1024, 1024, 1, 0, 40), ("k3 1024->1024", 1740, 1024, 3072, 3, 870, 16), ("qkv 1024->3072", 1740, 3072, 1024, 1, 0, 16),
                  ("integ 2048->1024", 1740, 1024, 2048, 1, 0, 20), ("decode fc M=256", 256, 4096, 1024, 1, 0, 40), ("clvp ff1", 51200, 3072, 768, 1, 0, 2)]
        for name, M, N, K, taps, seq, nw in shapes:
            for na in (1, 8):
                for xr in (1, 2, 4, 8, 0):
                    us = gemm_prod(M, N, K, taps, seq, nw=nw, na=na, xcd_rows=xr, chain=max(nw, 16))
                    print(f"xcd bands {name:18s} M={M} rotating A x{na} W x{nw}  xcd_rows={xr if xr else 'auto'}: {us:7.2f} us {tf(M, N, K, us):7.1f} TFLOP/s", flush=True)
    if "gemm2" in which:
        shapes = [("1x1 1024->1024", 1740, 1024, 1024, 1, 0), ("k3 1024->1024", 1740, 1024, 3072, 3, 870), ("qkv 1024->3072", 1740, 3072, 1024, 1, 0)]
        for name, M, N, K, taps, seq in shapes:
            us = gemm_prod(M, N, K, taps, seq)
            print(f"denoiser {name} M={M} PRODUCT gemm_launch (bias fetched before the k-loop): {us:7.2f} us {tf(M, N, K, us):7.1f} TFLOP/s", flush=True)
            for v in (0, 300, 400, 500, 501, 502, 200, 1300, 1301, 600, 1600):
                for pad in (0, 64):
                    us = gemm_exp(v, M, N, K, taps, seq, pad=pad)
                    label = EXTRA.get(v) or CFG[v // 100]
                    print(f"denoiser {name} M={M} exp {label:38s} row pad {pad:3d}: {us:7.2f} us {tf(M, N, K, us):7.1f} TFLOP/s", flush=True)
    if "gemm_large" in which:
        # the large-M users of the 128x128 product tile: CLVP speech tower (256 candidates x 200 codes), the conditioning-integrator
        # pre-pass (200 timesteps x 870 positions), a 15-chunk batched denoiser (15 x 2 x 870 rows)
        shapes = [("clvp qkv 768->2304", 51200, 2304, 768, 1, 0), ("clvp ff1 768->3072", 51200, 3072, 768, 1, 0), ("clvp ff2 1536->768", 51200, 768, 1536, 1, 0),
                  ("prepass 1x1 1024->1024", 174000, 1024, 1024, 1, 0), ("prepass k3 1024->1024", 174000, 1024, 3072, 3, 870),
                  ("batched denoiser 1x1", 26100, 1024, 1024, 1, 0), ("batched denoiser k3", 26100, 1024, 3072, 3, 870), ("batched denoiser qkv", 26100, 3072, 1024, 1, 0)]
        for name, M, N, K, taps, seq in shapes:
            us = gemm_prod(M, N, K, taps, seq, chain=4)
            print(f"large {name:24s} M={M} PRODUCT gemm_launch: {us:8.2f} us {tf(M, N, K, us):7.1f} TFLOP/s", flush=True)
            for v in (1600, 600, 2000, 2100, 2200, 2300, 2400):
                us = gemm_exp(v, M, N, K, taps, seq, chain=4)
                label = EXTRA.get(v) or CFG[v // 100]
                print(f"large {name:24s} M={M} exp {label:30s}: {us:8.2f} us {tf(M, N, K, us):7.1f} TFLOP/s", flush=True)
    if "gemm_denoiser" in which:
        shapes = [("1x1 1024->1024", 1740, 1024, 1024, 1, 0, 150), ("k3 1024->1024", 1740, 1024, 3072, 3, 870, 56), ("qkv 1024->3072", 1740, 3072, 1024, 1, 0, 56)]
        for name, M, N, K, taps, seq, nwc in shapes:
            for nw, tag in ((1, "hot W"), (nwc, "cold W")):
                us = gemm_prod(M, N, K, taps, seq, nw=nw)
                print(f"denoiser {name} M={M} PRODUCT gemm_launch ({tag}): {us:7.2f} us {tf(M, N, K, us):7.1f} TFLOP/s", flush=True)
                for cfg in (0, 1, 3, 9, 6, 7, 8, 4):
                    modes = (0, 1, 2) if cfg in (0, 1, 6) else (0,)
                    for mode in modes:
                        us = gemm_exp(cfg * 100 + mode, M, N, K, taps, seq, nw=nw)
                        print(f"denoiser {name} M={M} exp {CFG[cfg]:38s} {MODE[mode]:22s} ({tag}): {us:7.2f} us {tf(M, N, K, us):7.1f} TFLOP/s", flush=True)
                for v, label in ((200, "128x64 4w ring3 2 blocks/CU"), (500, "64x64 4w ring3 2 blocks/CU")):
                    us = gemm_exp(v, M, N, K, taps, seq, nw=nw)
                    print(f"denoiser {name} M={M} exp {label:38s} {'full':22s} ({tag}): {us:7.2f} us {tf(M, N, K, us):7.1f} TFLOP/s", flush=True)
    if "gemm_decode" in which:
        for name, N, K, sk in (("qkv", 3072, 1024, 1), ("proj", 1024, 1024, 4), ("fc", 4096, 1024, 1), ("proj2", 1024, 4096, 4), ("mel_head", 8194, 1024, 1)):
            nw = max(8, int(700e6 // (N * K * 2)))
            for M in (256, 32):
                for packed in (0, 1):
                    us = gemm_prod(M, N, K, splitk=sk, packed=packed, nw=nw)
                    print(f"decode {name:8s} M={M:3d} N={N} K={K} PRODUCT splitk={sk} packed={packed} cold W: {us:7.2f} us {N * K * 2 / us / 1e3:7.1f} GB/s weights", flush=True)
                if N % 64 == 0:
                    for v in (400, 500, 501):
                        for pad in (0, 64):
                            us = gemm_exp(v, M, N, K, nw=nw, pad=pad)
                            label = EXTRA.get(v) or CFG[v // 100]
                            print(f"decode {name:8s} M={M:3d} N={N} K={K} exp {label:38s} row pad {pad:3d} cold W: {us:7.2f} us {N * K * 2 / us / 1e3:7.1f} GB/s weights", flush=True)
                    us = gemm_exp(400, M, N, K, nw=1)
                    print(f"decode {name:8s} M={M:3d} N={N} K={K} exp {CFG[4]:38s} hot W: {us:7.2f} us", flush=True)
    if "bw2" in which:   # more waves / more loads in flight per lane than `bw`: where does the per-CU L2 -> CU rate saturate?
        for fp, tag in ((2 << 20, "2 MiB (one L2)"), (160 << 20, "160 MiB (Infinity Cache)")):
            for mode, mtag in ((0, "global_load_lds 16B"), (1, "global_load_dwordx4 -> VGPR, full lines")):
                for waves, wtag in ((8, "8 waves x 8 loads"), (16, "16 waves x 8 loads"), (108, "8 waves x 16 loads"), (116, "16 waves x 16 loads")):
                    for nb in (256, 512):
                        per = 1 << 20
                        us = bw_probe(mode, waves, nb, fp, per)
                        print(f"bw2 {tag:24s} {mtag:40s} {wtag:20s} x {nb} blocks: {us:8.2f} us  {nb * per / us / 1e6:7.2f} TB/s  {per * (nb / 256) / us / 1e3:6.1f} GB/s per CU", flush=True)
    if "kvpat" in which:  # the decode attention's K / V access pattern against contiguous streams of the same size and geometry (30 cold regions)
        lib.tt_kb_kv_pattern.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(D)]
        for tmax in (232, 128):
            for mode, tag in ((0, "product layout: K chunk-major 8 x 2 KB runs + V 16 KB"), (2, "K 16 KB + V 16 KB contiguous, two arrays"), (1, "one 32 KB run per (sequence, head)")):
                us = D(0)
                chk(lib.tt_kb_kv_pattern(mode, 256, 16, tmax, 30, 10, C.byref(us)))
                mb = 256 * 16 * 32768 / 1e6
                print(f"kvpat tmax {tmax:3d} {tag:60s}: {us.value:7.2f} us per launch of {mb:.0f} MB = {mb / us.value:5.2f} TB/s", flush=True)
    if "gemv" in which:  # round 6: a GEMV-shaped kernel against the product's skinny MFMA tile at the decode shapes, M = 1 .. 8 rows, cold weights, rotating A
        lib.tt_kb_gemv_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(D)]
        for name, N, K in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc", 4096, 1024), ("proj2", 1024, 4096), ("lm_head", 8196, 1024)):
            nw = max(8, int(700e6 // (N * K * 2)))
            for M in (1, 4, 8):
                if K == 4096 and M > 4:
                    continue
                up = gemm_prod(M, N, K, splitk=1, nw=nw, na=4)
                u = D(0)
                chk(lib.tt_kb_gemv_probe(M, N, K, nw, 4, 32, 10, C.byref(u)))
                print(f"gemv {name:8s} M={M} N={N} K={K}: product (32 x 16 MFMA tile, no split-K) {up:6.2f} us | GEMV probe {u.value:6.2f} us", flush=True)
    if "attn_line" in which:  # round 6: T(t) of the product decode attention against a pure-load kernel of its geometry, two bursts (K, then V) and one
        lib.tt_kb_kv_pattern2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(D)]
        for B in (256, 32):
            for t in (16, 32, 64, 96, 128):
                nl = t // 8
                by = (B * t) * 16 * 64 * 2 * 2
                row = [f"attn_line B={B:3d} own keys={t:3d} ({by / 1e6:6.1f} MB)"]
                us, _ = attn(0, B, 16, 59, t, 208, 30, chain=30)
                row.append(f"product kernel {us:6.2f} us")
                for ob, tag in ((0, "loads only, K burst then V burst"), (1, "loads only, ONE burst")):
                    u2 = D(0)
                    if ob and nl > 12:
                        row.append(f"{tag}: (register budget)")
                        continue
                    chk(lib.tt_kb_kv_pattern2(nl, ob, B, 16, 208, 30, 10, C.byref(u2)))
                    row.append(f"{tag} {u2.value:6.2f} us")
                print(" | ".join(row), flush=True)
    if "bw3" in which:   # working-set series: what does a set resident in the Infinity Cache (256 MiB) stream at, against one that only HBM holds?
        for mib in (16, 64, 128, 192, 512, 2048, 8192):
            for nb in (512, 1024):
                per = 1 << 20
                us = bw_probe(0, 8, nb, mib << 20, per)
                print(f"bw3 working set {mib:5d} MiB  global_load_lds 16B, 8 waves x 8 loads x {nb:4d} blocks of 1 MiB: {us:8.2f} us  {nb * per / us / 1e6:6.2f} TB/s", flush=True)
    if "ring" in which:  # ring depth / tile series at the decode shapes; rotating A (na = 4) like the decode step, cold W
        for name, N, K in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc", 4096, 1024), ("proj2", 1024, 4096)):
            nw = max(8, int(700e6 // (N * K * 2)))
            for M in (256, 32):
                sk = 4 if N == 1024 else 1
                us = gemm_prod(M, N, K, splitk=sk, nw=nw, na=4)
                print(f"ring {name:6s} M={M:3d} N={N} K={K} PRODUCT splitk={sk} rotating A x4, cold W: {us:7.2f} us", flush=True)
                vs = (400, 3500, 3000, 3100, 3400) + ((3200, 3300, 3600) if N == 1024 else ())
                for v in vs:
                    us = gemm_exp_na(v, M, N, K, nw=nw, na=4)
                    print(f"ring {name:6s} M={M:3d} N={N} K={K} exp {CFG[v // 100]:24s} rotating A x4, cold W: {us:7.2f} us", flush=True)
    if "gemm_ablate" in which:
        for name, N, K, sk in (("qkv", 3072, 1024, 1), ("fc", 4096, 1024, 1)):
            nw = max(8, int(700e6 // (N * K * 2)))
            for rep in range(2):
                us = gemm_prod(256, N, K, splitk=sk, nw=nw)
                print(f"ablate {name} M=256 PRODUCT (run-time outputs) cold W: {us:7.2f} us", flush=True)
                us = gemm_prod(256, N, K, splitk=sk, nw=nw, act=1)
                print(f"ablate {name} M=256 PRODUCT gelu -> T (compile-time outputs) cold W: {us:7.2f} us", flush=True)
                us = gemm_prod(256, N, K, splitk=4, nw=nw)
                print(f"ablate {name} M=256 PRODUCT split-K 4 slabs cold W: {us:7.2f} us", flush=True)
                for v, label in ((400, "exp 64x64"), (403, "exp 64x64 + bias quads behind the ring fill")):
                    us = gemm_exp(v, 256, N, K, nw=nw)
                    print(f"ablate {name} M=256 {label:46s} cold W: {us:7.2f} us", flush=True)
    if "attn" in which:
        for B in (256, 32):
            for tgen in (50, 100, 200):
                for v, label in ((10, "per-wave prefix kernel"), (11, "shared prefix in LDS, 16 seq / workgroup"), (12, "shared prefix in LDS, 4 seq / workgroup"), (0, "product (auto)")):
                    us, md = attn(v, B, 16, 59, tgen, 202, 8)
                    by = (B * tgen + 59) * 16 * 64 * 2 * 2
                    print(f"decode_attn B={B} tgen={tgen} {label:36s}: {us:7.2f} us {by / us / 1e3:7.1f} GB/s  max|diff| vs per-wave kernel {md:.3e}", flush=True)


if __name__ == "__main__":
    main()
