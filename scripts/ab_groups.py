#!/usr/bin/env python
"""AR stage with several utterances in one decode batch (tt_ar_prefill_group) at the benchmark shape: ms per decode step and per
utterance for G = 1, 2, 4, 8 groups of 256 candidates x 200 tokens.  `python scripts/ab_groups.py 1 2 4 8`"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    groups = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    from tortoise_tts_amd import stages, weights as W
    from tortoise_tts_amd.config import ARConfig
    cfg = ARConfig()
    sd = W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(cfg), 1234), cfg)
    N, M = 256, 200
    Gmax = max(groups)
    ar = stages.ArStage(sd, cfg, max_batch=N * Gmax, max_new_tokens=M, max_latent_candidates=1, max_groups=Gmax)
    g = torch.Generator().manual_seed(4)
    utts = [(torch.randn(1, 1024, generator=g).cuda() * 0.5, F.pad(torch.randint(1, 255, (1, int(torch.randint(60, 111, (1,), generator=g))), generator=g).int(), (0, 1)).cuda())
            for _ in range(Gmax)]
    with torch.no_grad():
        for G in groups:
            times = []
            for r in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for gi in range(G):
                    ar.prefill_group(gi, G, *utts[gi])
                codes, n = ar.generate(N * G, M, seed=77)
                torch.cuda.synchronize()
                if r:
                    times.append(time.perf_counter() - t0)
            t = sum(times) / len(times)
            import hashlib
            dig = hashlib.sha256(codes.cpu().numpy().tobytes()).hexdigest()[:12]
            print("%-8s groups %2d x %d candidates x %d tokens: %.1f ms total, %.3f ms/step, %.1f ms per utterance  codes %s" %
                  (os.environ.get("AB_NAME", "base"), G, N, n, 1e3 * t, 1e3 * t / n, 1e3 * t / G, dig), flush=True)


if __name__ == "__main__":
    main()
