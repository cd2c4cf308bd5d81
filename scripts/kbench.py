#!/usr/bin/env python
"""Micro-benchmarks of the hot kernels at the shapes the 'standard' preset produces (run on the MI355X).
Prints one line per case: average microseconds over graph-free back-to-back launches, TFLOP/s and GB/s."""
import math
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tortoise_tts_amd import engine as E  # noqa: E402

lib = E.init()
T = torch.bfloat16
DT = E.TT_BF16


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters  # us


def gemm_case(name, M, N, K, taps=1, seq=0, splitk=1):
    A = torch.randn(M, K // taps, device="cuda").to(T)
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(T)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(max(splitk, 1), M, N, device="cuda")
    us = timeit(lambda: E.check(lib.tt_op_gemm(DT, E.ptr(A), K // taps, E.ptr(W), K, M, N, K, taps, seq, splitk,
                                               E.ptr(bias) if splitk == 1 else None, 0, None, E.ptr(out), None, None)))
    fl = 2.0 * M * N * K
    by = (N * K + M * K / taps) * 2 + M * N * 4 * splitk
    print(f"gemm {name:28s} M={M:5d} N={N:5d} K={K:5d} sk={splitk}: {us:8.2f} us  {fl / us / 1e6:8.1f} TFLOP/s  {by / us / 1e3:8.1f} GB/s")


def main():
    which = sys.argv[1:] or ["gemm", "gn", "flash", "ln"]
    if "gemm" in which:
        for B in (256, 32):
            gemm_case(f"decode qkv B={B}", B, 3072, 1024)
            gemm_case(f"decode fc B={B}", B, 4096, 1024)
            for sk in (1, 2, 4, 8):
                gemm_case(f"decode proj B={B}", B, 1024, 1024, splitk=sk)
                gemm_case(f"decode proj2 B={B}", B, 1024, 4096, splitk=sk)
            gemm_case(f"mel_head B={B}", B, 8194, 1024)
        gemm_case("diff 1x1 S=870x2", 1740, 1024, 1024)
        gemm_case("diff k3 S=870x2", 1740, 1024, 3072, taps=3, seq=870)
        gemm_case("diff qkv S=870x2", 1740, 3072, 1024)
        gemm_case("diff integ S=870x2", 1740, 1024, 2048)
        gemm_case("diff 1x1 S=2176x2", 4352, 1024, 1024)
        gemm_case("clvp ff1 256x200", 51200, 3072, 768)
        gemm_case("clvp out 256x200", 51200, 768, 768)
        gemm_case("square 4096", 4096, 4096, 4096)
    if "tiles" in which:
        for M, N, K, taps, seq in ((1740, 1024, 1024, 1, 0), (1740, 1024, 3072, 3, 870), (1740, 3072, 1024, 1, 0), (256, 4096, 1024, 1, 0),
                                   (256, 1024, 4096, 1, 0), (4096, 4096, 4096, 1, 0)):
            gemm_case(f"tile={os.environ.get('TT_GEMM_TILE', 'auto')}", M, N, K, taps=taps, seq=seq)
    if "cold" in which:
        # decode GEMMs with HBM-cold weights: cycle through enough distinct weight matrices to defeat the 256 MB Infinity Cache
        for (name, M, N, K, sk) in (("qkv", 256, 3072, 1024, 1), ("fc", 256, 4096, 1024, 1), ("proj", 256, 1024, 1024, 4),
                                    ("proj2", 256, 1024, 4096, 4), ("qkv32", 32, 3072, 1024, 1), ("fc32", 32, 4096, 1024, 1)):
            nW = max(8, int(600e6 // (N * K * 2)))
            Ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).to(T) for _ in range(nW)]
            A = torch.randn(M, K, device="cuda").to(T)
            bias = torch.randn(N, device="cuda")
            out = torch.zeros(max(sk, 1), M, N, device="cuda")
            it = [0]

            def fn():
                W = Ws[it[0] % nW]
                it[0] += 1
                E.check(lib.tt_op_gemm(DT, E.ptr(A), K, E.ptr(W), K, M, N, K, 1, 0, sk, E.ptr(bias) if sk == 1 else None, 0, None,
                                       E.ptr(out), None, None))
            us = timeit(fn, iters=4 * nW, warm=nW)
            print(f"cold gemm {name:6s} tile={os.environ.get('TT_GEMM_TILE', 'auto'):4s} M={M} N={N} K={K} sk={sk}: {us:8.2f} us  {N * K * 2 / us / 1e3:8.1f} GB/s weights")
            del Ws
    if "coldpacked" in which:
        from tortoise_tts_amd.pack import Holder
        hold = Holder(torch.device("cuda"), DT)
        for (name, M, N, K, sk) in (("qkv", 256, 3072, 1024, 1), ("fc", 256, 4096, 1024, 1), ("proj", 256, 1024, 1024, 4),
                                    ("proj2", 256, 1024, 4096, 4), ("fc32", 32, 4096, 1024, 1)):
            nW = max(8, int(600e6 // (N * K * 2)))
            Ws = []
            for _ in range(nW):
                hold.keep.clear()
                Ws.append(hold.op_packed(torch.randn(N, K, device="cuda") / math.sqrt(K)))
            A = torch.randn(M, K, device="cuda").to(T)
            out = torch.zeros(max(sk, 1), M, N, device="cuda")
            it = [0]

            def fn():
                W = Ws[it[0] % nW]
                it[0] += 1
                E.check(lib.tt_op_gemm_packed(DT, E.ptr(A), K, E.ptr(W), M, N, K, sk, None, 0, None, E.ptr(out), None, None))
            us = timeit(fn, iters=4 * nW, warm=nW)
            print(f"cold packed gemm {name:6s} M={M} N={N} K={K} sk={sk}: {us:8.2f} us  {N * K * 2 / us / 1e3:8.1f} GB/s weights")
            del Ws
    if "kscale" in which:
        for M in (32, 256):
            for K in (64, 256, 1024, 4096):
                N = 4096
                nW = max(8, int(600e6 // (N * K * 2)))
                nW = min(nW, 256)
                Ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).to(T) for _ in range(nW)]
                A = torch.randn(M, K, device="cuda").to(T)
                out_t = torch.zeros(M, N, device="cuda", dtype=T)
                it = [0]

                def fn():
                    W = Ws[it[0] % nW]
                    it[0] += 1
                    E.check(lib.tt_op_gemm(DT, E.ptr(A), K, E.ptr(W), K, M, N, K, 1, 0, 1, None, 0, None, None, E.ptr(out_t), None))
                us = timeit(fn, iters=4 * nW, warm=nW)
                print(f"kscale M={M} N={N} K={K}: {us:8.2f} us  ({N * K * 2 / 1e6:.1f} MB weights, {N * K * 2 / us / 1e3:8.1f} GB/s)")
                del Ws
    if "gn" in which:
        for (B, S, C_) in ((2, 870, 1024), (2, 2176, 1024)):
            x = torch.randn(B, S, C_, device="cuda")
            g, b = torch.randn(C_, device="cuda"), torch.randn(C_, device="cuda")
            ws = torch.zeros(lib.tt_op_groupnorm_workspace(B, S) // 4 + 16, device="cuda")
            o = torch.zeros(B, S, C_, device="cuda", dtype=T)
            us = timeit(lambda: E.check(lib.tt_op_groupnorm(DT, E.ptr(x), B, S, C_, E.ptr(g), E.ptr(b), None, E.ACT_SILU, E.ptr(o), None,
                                                            E.ptr(ws), None)))
            print(f"groupnorm B={B} S={S} C={C_}: {us:8.2f} us  {B * S * C_ * 10 / us / 1e3:8.1f} GB/s")
    if "ln" in which:
        for M in (256, 1740):
            x = torch.randn(M, 1024, device="cuda")
            g, b = torch.randn(1024, device="cuda"), torch.randn(1024, device="cuda")
            o = torch.zeros(M, 1024, device="cuda", dtype=T)
            us = timeit(lambda: E.check(lib.tt_op_layernorm(DT, E.ptr(x), M, 1024, E.ptr(g), E.ptr(b), 1e-5, 0, E.ptr(o), None, None)))
            print(f"layernorm M={M}: {us:8.2f} us")
    if "flash" in which:
        for (B, H, n, causal) in ((2, 16, 870, 0), (2, 16, 2176, 0), (256, 12, 200, 0), (1, 16, 260, 1)):
            n_pad = (n + 31) // 32 * 32
            q = (torch.randn(B, H, n, 64, device="cuda") * 0.2).to(T)
            k = torch.randn(B, H, n, 64, device="cuda").to(T)
            vt = torch.randn(B, H, 64, n_pad, device="cuda").to(T)
            rp = torch.randn(H, 129, device="cuda")
            o = torch.zeros(B, n, H * 64, device="cuda", dtype=T)
            us = timeit(lambda: E.check(lib.tt_op_flash_attention(DT, E.ptr(q), E.ptr(k), E.ptr(vt), E.ptr(o), B, H, n, n_pad, causal,
                                                                  E.ptr(rp) if not causal else None, None)))
            fl = 4.0 * B * H * n * n * 64 * (0.5 if causal else 1.0)
            print(f"flash B={B} H={H} n={n} causal={causal}: {us:8.2f} us  {fl / us / 1e6:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
