"""In-situ A/B of the 256 x 256 GEMM tile kernels (ttx_kernel_variant(TTX_GEMM_P8): 1 = 8 waves / eight phases, csrc/gemm_p8.h; 0 = 16 waves / two stages) through
tt_op_gemm at the large-M product shapes: chains of launches between two events, variants alternating in one process."""
import math
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tortoise_tts_amd import engine as E


def main():
    lib = E.init()
    g = torch.Generator().manual_seed(3)
    shapes = [("batched denoiser 1x1 -> f32 + skip", 26100, 1024, 1024, "res"), ("batched denoiser qkv-like -> T", 26100, 3072, 1024, "t"),
              ("pre-pass chunk 1x1 -> f32 + skip", 27840, 1024, 1024, "res"), ("clvp ff1-like gelu -> T", 51200, 3072, 768, "gelu"),
              ("square 8192 -> T", 8192, 8192, 8192, "t")]
    quick = "--quick" in sys.argv
    if "--prepass" in sys.argv:  # the conditioning-integrator pre-pass chunks: 18 (benchmark) and 24 (the PMC pass) timesteps x 2 rows x 870 positions
        shapes = [("pre-pass 18 steps qkv-like -> T", 31320, 3072, 1024, "t"), ("pre-pass 24 steps qkv-like -> T", 41760, 3072, 1024, "t"),
                  ("pre-pass 18 steps 1x1 -> f32 + skip", 31320, 1024, 1024, "res"), ("pre-pass 24 steps 1x1 -> f32 + skip", 41760, 1024, 1024, "res"),
                  ("clvp 256 x 200 linear -> T", 51200, 768, 768, "t"), ("clvp qkv-like -> T", 51200, 2304, 768, "t")]
    for dt, tdt, name in ((E.TT_F16, torch.float16, "fp16"),) if quick else ((E.TT_F16, torch.float16, "fp16"), (E.TT_BF16, torch.bfloat16, "bf16")):
        for label, M, N, K, form in shapes:
            A = torch.randn(M, K, generator=g).to(tdt).cuda()
            Wt = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(tdt).cuda()
            bias = torch.randn(N, generator=g).cuda()
            o32 = torch.randn(M, N, generator=g).cuda() if form == "res" else None
            ot = torch.zeros(M, N, device="cuda", dtype=tdt) if form != "res" else None

            def once():
                if form == "res":
                    E.check(lib.tt_op_gemm(dt, E.ptr(A), K, E.ptr(Wt), K, M, N, K, 1, 0, 1, E.ptr(bias), E.ACT_NONE, E.ptr(o32), E.ptr(o32), None, None))
                else:
                    E.check(lib.tt_op_gemm(dt, E.ptr(A), K, E.ptr(Wt), K, M, N, K, 1, 0, 1, E.ptr(bias), E.ACT_GELU_TANH if form == "gelu" else E.ACT_NONE, None, None, E.ptr(ot), None))

            res = {}
            for rnd in range(3):
                for v in (1, 0):
                    lib.ttx_kernel_variant(E.TTX_GEMM_P8, v)
                    once()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    n = 20
                    e0.record()
                    for _ in range(n):
                        once()
                    e1.record()
                    torch.cuda.synchronize()
                    res.setdefault(v, []).append(e0.elapsed_time(e1) * 1e3 / n)
            lib.ttx_kernel_variant(E.TTX_GEMM_P8, 1)
            us1, us0 = min(res[1]), min(res[0])
            tf = lambda us: 2.0 * M * N * K / us / 1e6
            print("ab gemm %-4s %-36s M=%6d N=%5d K=%5d: eight-phase %8.1f us %7.1f TFLOP/s | 16-wave %8.1f us %7.1f TFLOP/s | x%.3f" %
                  (name, label, M, N, K, us1, tf(us1), us0, tf(us0), us0 / us1), flush=True)


if __name__ == "__main__":
    main()
