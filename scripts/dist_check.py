"""End-to-end sharding check: the same seeded utterance rendered by 1 rank and by R ranks must give the same audio.
Sampled codes do not depend on the sharding (Philox keyed by the global candidate index), so the winner is the same
candidate; with >= 2 ranks its diffusion tail is split over ranks 0/1, which differs from the batched tail only by
summation grouping in the GroupNorm statistics.

    python scripts/dist_check.py --out gpurun_out/wav_r1.pt
    TT_DIST_SHARE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        scripts/dist_check.py --out gpurun_out/wav_r2.pt          # 1-GPU box: ranks share the device, gloo
    python scripts/dist_check.py --compare gpurun_out/wav_r1.pt gpurun_out/wav_r2.pt
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--compare", nargs=2)
    ap.add_argument("--candidates", type=int, default=16)
    ap.add_argument("--mel-tokens", type=int, default=60)
    ap.add_argument("--iterations", type=int, default=30)
    args = ap.parse_args()
    if args.compare:
        a, b = (torch.load(p) for p in args.compare)
        assert torch.equal(a["codes"], b["codes"]), "the ranked winner differs between shardings"
        rel = float((a["wav"] - b["wav"]).norm() / a["wav"].norm())
        print(f"[parity] winner codes identical; waveform rel_l2 1-rank vs {b['world']}-rank = {rel:.3e} (tol 2.0e-02)")
        assert rel < 2e-2
        return
    from bench import synthetic_weights, synthetic_prompt
    from tortoise_tts_amd import dist as tdist
    rank, world, _ = tdist.init_from_env_or_exit()
    from tortoise_tts_amd.api import TextToSpeech
    text, latents = synthetic_prompt()
    tts = TextToSpeech(state_dicts=synthetic_weights(), max_candidates=args.candidates // world, max_mel_tokens=max(args.mel_tokens, 32))
    wav = tts.tts(text, conditioning_latents=latents, k=1, verbose=False, use_deterministic_seed=77,
                  num_autoregressive_samples=args.candidates, max_mel_tokens=args.mel_tokens,
                  diffusion_iterations=args.iterations, cond_free=True)
    if rank == 0:
        torch.save({"wav": wav.float().cpu(), "codes": tts.last_best_codes.cpu(), "world": world, "split": tts.split_diffusion}, args.out)
        print(f"rank 0 of {world}: wav {tuple(wav.shape)} split_diffusion={tts.split_diffusion} stages={tts.timings}")
    tdist.barrier()


if __name__ == "__main__":
    main()
