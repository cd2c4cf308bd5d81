"""Time one denoiser row (B = 1, what each of the two GPUs runs per step in the split tail) against the batched
cond+uncond step (B = 2, the single-GPU tail) at the 'standard' shape (S = 870)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tortoise_tts_amd import stages, weights as W  # noqa: E402
from tortoise_tts_amd.config import DiffusionConfig  # noqa: E402
from tortoise_tts_amd.schedule import Schedule  # noqa: E402

cfg = DiffusionConfig()
sd = W.synthetic_state_dict(W.diffusion_manifest(cfg), 3)
M = 200
S = M * 4 * 24000 // 22050
g = torch.Generator().manual_seed(0)
st = stages.DiffusionStage(sd, cfg, max_seq=S + 8, max_codes=M + 8, max_steps=64)
st.condition(torch.randn(1, M, 1024, generator=g), torch.randn(1, 2048, generator=g) * 0.5, S)
N = 40
x, noise = torch.randn(1, 100, S, generator=g), torch.randn(N, 1, 100, S, generator=g)
sched = Schedule(N, 4000, True, 2.0)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.sample(sched, x, noise)
    torch.cuda.synchronize(); t_b2 = (time.perf_counter() - t0) / N
for _ in range(2):
    st.split_begin(sched, x, noise, 0)
    rows = torch.zeros(2, S, cfg.out_channels, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        rows[0].copy_(st.split_forward())
        st.split_update(rows)
    torch.cuda.synchronize(); t_b1 = (time.perf_counter() - t0) / N
    st.split_end()
print(f"denoiser step S={S}: batched cond+uncond {1e3 * t_b2:.3f} ms, one row + update {1e3 * t_b1:.3f} ms (tile override: {os.environ.get('TT_GEMM_TILE', 'auto')})")
