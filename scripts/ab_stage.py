#!/usr/bin/env python
"""In-situ A/B timing of ONE stage at the benchmark shape ('standard': 256 candidates x 200 mel tokens / 200 diffusion
iterations at S = 870), without the rest of the pipeline: `TORTOISE_MI355X_LIB=<alt .so> python scripts/ab_stage.py ar diff`.
Both builds of an A/B must run inside the same gpurun call (boxes differ by +-3 %).  Prints one line per stage:
    ab <tag> ar   : ms per decode step (mean of R repetitions of the full 200-token generation, graph replay), min
    ab <tag> diff : ms per sampler iteration
and a checksum of the produced codes / mel so that bit-identity between two builds can be read off the log."""
import argparse
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stages", nargs="*", default=["ar", "diff"])
    ap.add_argument("--tag", default=os.environ.get("AB_TAG", "base"))
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--candidates", type=int, default=256)
    ap.add_argument("--mel-tokens", type=int, default=200)
    ap.add_argument("--iterations", type=int, default=200)
    ap.add_argument("--ar-variants", default="",
                    help="';'-separated host-lookahead settings of tt_ar_set_option to time one after the other on ONE handle")
    ap.add_argument("--typical-variants", default="",
                    help="';'-separated typical_mass settings (0 = off) of the AR stage's sampler, timed one after the other on ONE handle, e.g. '0;0.9;0;0.9'")
    ap.add_argument("--flash-variants", default="",
                    help="';'-separated ttx_kernel_variant(TTX_FLASH32) settings (1 = 32-query-wave attention kernel, 0 = 16-query waves) to time the diffusion "
                         "stage with, one fresh stage object each, e.g. '1;0;1;0'")
    ap.add_argument("--gn-variants", default="",
                    help="';'-separated TT_DIFF_OPT_FUSED_GN values (0 = stand-alone applies, 1 = ResBlock in_layers fused, 2 = + the attention norm on "
                         "the QKV GEMM's A path) timed one after the other on ONE diffusion stage object, e.g. '1;2;1;2'")
    ap.add_argument("--voc-variants", default="", help="';'-separated ttx_kernel_variant(TTX_VOC_MFMA) settings (1 = f32-MFMA audio-rate kernels, 0 = VALU) for the 'voc' stage")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    args = ap.parse_args()
    from bench import bench_prompt
    from tortoise_tts_amd import stages, weights as W
    from tortoise_tts_amd.config import ARConfig, DiffusionConfig
    from tortoise_tts_amd.schedule import Schedule
    import torch.nn.functional as F
    text, (auto, diffc) = bench_prompt()
    dev = "cuda"
    with torch.no_grad():
        if "ar" in args.stages:
            cfg = ARConfig()
            sd = W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(cfg), 1234), cfg)
            from tortoise_tts_amd import engine as E
            dt = E.dtype_code(args.dtype)
            ar = stages.ArStage(sd, cfg, dtype=dt, max_batch=args.candidates, max_new_tokens=max(args.mel_tokens, 32), max_latent_candidates=1)
            tt = F.pad(text.int()[None], (0, 1)).to(dev)
            variants = [tuple(int(v) for v in item.split(",")) for item in args.ar_variants.split(";") if item.strip()] or [None]
            masses = [float(v) for v in args.typical_variants.split(";") if v.strip()]
            if masses:
                variants = [("typical", m) for m in masses]
            for var in variants:
                tag = args.tag
                mass = 0.0
                if var is not None and var[0] == "typical":
                    mass = var[1]
                    tag = "typical=%g" % mass
                elif var is not None:
                    ar.set_option(E.TT_AR_OPT_LOOKAHEAD, var[0])
                    tag = "look=%d" % var[0]
                times = []
                codes = None
                for r in range(args.reps + 1):
                    ar.prefill(auto.to(dev), tt)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    codes, n = ar.generate(args.candidates, args.mel_tokens, seed=77, typical_mass=mass)
                    torch.cuda.synchronize()
                    if r:
                        times.append((time.perf_counter() - t0) / n)
                print("ab %-12s ar   B=%d n=%d: %.4f ms/step (min %.4f)  total %.1f ms  codes %s  drains %d  launches/step %d" %
                      (tag, args.candidates, n, 1e3 * sum(times) / len(times), 1e3 * min(times), 1e3 * n * sum(times) / len(times), digest(codes),
                       ar.stat(1), ar.stat(2)), flush=True)
            ar.close()
            del ar
        if "diff" in args.stages:
            cfg = DiffusionConfig()
            sd = W.synthetic_state_dict(W.diffusion_manifest(cfg), 1236)
            M = args.mel_tokens
            S = M * 4 * 24000 // 22050
            from tortoise_tts_amd import engine as E
            g = torch.Generator().manual_seed(5)
            lat = torch.randn(1, M, 1024, generator=g).to(dev)
            sched = Schedule(args.iterations, cfg.trained_steps, True, 2)
            x = torch.randn(1, 100, S, generator=g).to(dev)
            noise = torch.randn(args.iterations, 1, 100, S, generator=g).to(dev)
            fvs = [int(v) for v in args.flash_variants.split(";") if v.strip()] or [None]
            for fv in fvs:
                tag = args.tag
                if fv is not None:
                    E.load_library().ttx_kernel_variant(E.TTX_FLASH32, fv)
                    tag = "flash32=%d" % fv
                df = stages.DiffusionStage(sd, cfg, dtype=E.dtype_code(args.dtype), max_seq=max(S, 128), max_codes=max(M, 64), max_steps=args.iterations)
                for gv in [int(v) for v in args.gn_variants.split(";") if v.strip()] or [None]:
                    gtag = tag
                    if gv is not None:
                        df.set_option(E.TT_DIFF_OPT_FUSED_GN, gv)
                        gtag = "fused_gn=%d" % gv
                    times = []
                    mel = None
                    for r in range(args.reps + 1):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        df.condition(lat, diffc.to(dev), S)
                        mel = df.sample(sched, x, noise)
                        torch.cuda.synchronize()
                        if r:
                            times.append((time.perf_counter() - t0) / args.iterations)
                    print("ab %-10s diff S=%d it=%d: %.4f ms/iteration (min %.4f)  total %.1f ms  mel %s" %
                          (gtag, S, args.iterations, 1e3 * sum(times) / len(times), 1e3 * min(times), 1e3 * args.iterations * sum(times) / len(times), digest(mel)), flush=True)
                df.close()
            if fvs != [None]:
                E.load_library().ttx_kernel_variant(E.TTX_FLASH32, 1)
        if "voc" in args.stages:
            from tortoise_tts_amd import engine as E
            from tortoise_tts_amd.config import VocoderConfig
            vcfg = VocoderConfig()
            vsd = W.fold_weight_norm(W.synthetic_state_dict(W.vocoder_manifest(vcfg), 1234 + 3))
            S = args.mel_tokens * 4 * 24000 // 22050
            g = torch.Generator().manual_seed(6)
            mel = (torch.randn(1, 100, S, generator=g) * 2 - 5).to(dev)
            z = torch.randn(1, vcfg.noise_dim, S + 10, generator=g).to(dev)
            vs = stages.VocoderStage(vsd, vcfg, dtype=E.dtype_code(args.dtype), max_frames=S + 8)
            for vv in [int(v) for v in args.voc_variants.split(";") if v.strip()] or [1]:
                E.load_library().ttx_kernel_variant(E.TTX_VOC_MFMA, vv)
                wav = vs.inference(mel, z)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    wav = vs.inference(mel, z)
                torch.cuda.synchronize()
                print("ab voc_mfma=%d voc  S=%d: %.3f ms per utterance  wav %s  guard %d" % (vv, S, 1e3 * (time.perf_counter() - t0) / 20, digest(wav), vs.guard()), flush=True)
            E.load_library().ttx_kernel_variant(E.TTX_VOC_MFMA, 1)
            vs.close()


if __name__ == "__main__":
    main()
