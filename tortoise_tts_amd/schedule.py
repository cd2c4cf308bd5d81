"""Diffusion schedule tables for the engine's sampler.

Restates what `load_discrete_vocoder_diffuser` builds (reference: tortoise/api.py:64-70 ->
SpacedDiffusion(use_timesteps=space_timesteps(4000, [steps]), model_mean_type='epsilon',
model_var_type='learned_range', betas=get_named_beta_schedule('linear', 4000), conditioning_free,
conditioning_free_k); tortoise/utils/diffusion.py:94-111, 192-249, 1102-1116, 1152-1205).  Tables are
float64 like the reference's numpy arrays; the engine receives per-step float32 scalars, which is
what `_extract_into_tensor` (diffusion.py:1237-1250) hands the reference's fp32 tensors.
"""
import numpy as np


def space_timesteps(num_timesteps, section_counts):
    """diffusion.py:1152-1205 for integer section counts (api.py:68 passes [desired_steps])."""
    if isinstance(section_counts, int):
        section_counts = [section_counts]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start_idx = 0
    all_steps = []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac_stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            all_steps.append(start_idx + round(cur))
            cur += frac_stride
        start_idx += size
    return set(all_steps)


class Schedule:
    def __init__(self, steps, trained_steps=4000, cond_free=True, cond_free_k=2.0):
        scale = 1000 / trained_steps
        base_betas = np.linspace(scale * 0.0001, scale * 0.02, trained_steps, dtype=np.float64)
        base_alphas_cumprod = np.cumprod(1.0 - base_betas, axis=0)
        use = space_timesteps(trained_steps, [steps])
        last = 1.0
        betas, tmap = [], []
        for i, ac in enumerate(base_alphas_cumprod):
            if i in use:
                betas.append(1 - ac / last)
                last = ac
                tmap.append(i)
        betas = np.array(betas, dtype=np.float64)
        self.timestep_map = np.array(tmap, dtype=np.int64)
        self.num_timesteps = int(betas.shape[0])
        self.cond_free = bool(cond_free)
        self.cond_free_k = float(cond_free_k)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.betas = betas
        self.sqrt_recip_ac = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_ac = np.sqrt(1.0 / ac - 1)
        posterior_variance = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.post_logvar_clipped = np.log(np.append(posterior_variance[1], posterior_variance[1:]))
        self.log_betas = np.log(betas)
        self.coef1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
        self.coef2 = (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)

    @staticmethod
    def f32(arr, i):
        return float(np.float32(arr[i]))
