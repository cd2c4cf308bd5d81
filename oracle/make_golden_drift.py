"""TEST INFRASTRUCTURE ONLY — full-WIDTH sampler goldens from the REFERENCE's own DiffusionTts + SpacedDiffusion.

Run in the build container (needs /root/reference; ~20 minutes of CPU on 8 cores):   python -m oracle.make_golden_drift
Writes tests/golden/full_drift.npz: the final x0 of the reference's `SpacedDiffusion.p_sample_loop`
(tortoise/utils/diffusion.py:533-621) on the BENCHMARKED denoiser (1024 channels, 10 layers, S = 870, bench.py's synthetic
weights and prompt) over
  * the 'standard' schedule          — 200 iterations, conditioning-free guidance on (api.py:327),
  * the 'high_quality' schedule      — 400 iterations, conditioning-free guidance on (api.py:328),
  * the 'ultra_fast' schedule        — 30 iterations, cond_free=False (api.py:325; diffusion.py:341-384 takes the plain branch),
  * the 'fast' schedule (round 6)    — 80 iterations, conditioning-free guidance on (api.py:326: BASELINE config #2's own spaced schedule,
                                       `space_timesteps(4000, [80])`, diffusion.py:1152-1205),
with injected noise that the GPU tests rebuild from the seeds below (only the 348 KB outputs are stored).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402
from oracle import make_golden as G  # noqa: E402
from oracle import make_golden_full as GF  # noqa: E402
from tortoise_tts_amd.config import DiffusionConfig  # noqa: E402

OUT = G.OUT
# (name, iterations, cond_free, seed) — the GPU tests rebuild (latents, x_T, per-step noise) with GF.diff_inputs(cfg, 200, seed, N)
CASES = (("std200", 200, True, 41), ("hq400", 400, True, 42), ("uf30", 30, False, 43), ("fast80", 80, True, 44))


def ref_loop(ref, m, N, S, x, code_emb, step_noise, cond_free):
    """SpacedDiffusion.p_sample_loop with the per-step randn_like replaced by the injected tensors (diffusion.py:522)."""
    diffuser = ref.SpacedDiffusion(use_timesteps=ref.space_timesteps(4000, [N]), model_mean_type='epsilon',
                                   model_var_type='learned_range', loss_type='mse',
                                   betas=ref.get_named_beta_schedule('linear', 4000), conditioning_free=cond_free,
                                   conditioning_free_k=2.0)
    import tortoise.utils.diffusion as rd
    order = list(reversed(range(N)))
    calls = {"n": 0}
    orig = rd.th.randn_like

    def fake_randn_like(t):
        i = order[calls["n"]]
        calls["n"] += 1
        return step_noise[i]
    rd.th.randn_like = fake_randn_like
    try:
        return diffuser.p_sample_loop(m, (1, 100, S), noise=x.clone(), model_kwargs={'precomputed_aligned_embeddings': code_emb},
                                      progress=False)
    finally:
        rd.th.randn_like = orig


@torch.no_grad()
def main():
    import bench
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", os.cpu_count() or 1)))
    ref = ref_shims.import_reference()
    sds = bench.synthetic_weights()
    cfg = DiffusionConfig()
    m = GF.build_ref_diffusion(ref, cfg, sds["diffusion"])
    _, _, cond = GF.prompt()
    path = os.path.join(OUT, "full_drift.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}  # cases already generated are kept (the 400-iteration one takes ~15 min)
    for name, N, cond_free, seed in CASES:
        if name in out:
            continue
        t0 = time.time()
        S, latents, x, step_noise = GF.diff_inputs(cfg, M=GF.DIFF_M, seed=seed, steps=N)
        code_emb = m.timestep_independent(latents, cond, S, False)
        out[name] = ref_loop(ref, m, N, S, x, code_emb, step_noise, cond_free).numpy()
        print(f"{name}: {N} iterations cond_free={cond_free} in {time.time() - t0:.0f} s", flush=True)
        np.savez_compressed(os.path.join(OUT, "full_drift.npz"), **out)


if __name__ == "__main__":
    main()
