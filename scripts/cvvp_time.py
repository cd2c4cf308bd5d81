"""CVVP scoring (tts(cvvp_amount > 0), csrc/cvvp.hip) at the benchmark's candidate shape: ms per tt_cvvp_score call (scripts/gpu.sh py)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tortoise_tts_amd import stages, weights as W, engine as E
from tortoise_tts_amd.config import CVVPConfig, CLVPConfig
cfg = CVVPConfig()
sd = W.synthetic_state_dict(W.cvvp_manifest(cfg), seed=31)
g = torch.Generator().manual_seed(1)
mels = (torch.randn(1, 2, 80, 517, generator=g) * 2 - 5).cuda()          # device-resident, as inside tts()
codes = torch.randint(0, 8192, (256, 200), generator=g).int().cuda()
for name, dt in (("fp16", E.TT_F16), ("bf16", E.TT_BF16)):
    st = stages.CvvpStage(sd, cfg, dtype=dt, max_rows=256 * 200, max_cond_frames=520)
    st.score(mels, codes); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        s = st.score(mels, codes)
    torch.cuda.synchronize()
    dt_ms = (time.perf_counter() - t0) / 10 * 1e3
    fl = 2 * 256 * 200 * (8 * (4 * 512 * 512 + 3 * 512 * 512) + 5 * 512 * 512) + 8 * 4 * 256 * 200 * 200 * 512 + 4 * 256 * 200 * 200 * 512
    print(f"cvvp_time {name}: 256 candidates x 200 codes, 2 clips x 517 frames: {dt_ms:.3f} ms per call ({fl / dt_ms / 1e9:.0f} TFLOP/s of {fl / 1e12:.2f} TFLOP)", flush=True)
    st.close()
