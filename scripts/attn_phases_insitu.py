#!/usr/bin/env python
"""Phase stamps of the LAST decode-attention launch of a real generation (layer 30 of the last step, graph replay, cold KV cache: the
30 layers' caches are 6 GB at 256 candidates) - the in-situ counterpart of scripts/attn_phases.py, whose repeated launches on one
105 MB cache are partly served by the 256 MB Infinity Cache.
    TORTOISE_MI355X_LIB=tortoise_tts_amd/lib/libtortoise_mi355x_astamps.so python scripts/attn_phases_insitu.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from bench import bench_prompt  # noqa: E402
from tortoise_tts_amd import engine as E, stages, weights as W  # noqa: E402
from tortoise_tts_amd.config import ARConfig  # noqa: E402

PH = ["entry -> loads requested", "wait: staged prefix, barrier, query", "prefix scores (LDS)", "own scores (K stream)", "first V rows requested, softmax",
      "prefix PV (LDS)", "own PV (V stream)", "reduce + store"]
lib = E.init()
lib.ttx_attn_stamps.restype = C.c_int
lib.ttx_attn_stamps.argtypes = [C.c_void_p, C.c_int]
cfg = ARConfig()
sd = W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(cfg), 1234), cfg)
B = 256
cap = int(os.environ.get("KV_CAPACITY", "200"))
ar = stages.ArStage(sd, cfg, dtype=E.TT_F16, max_batch=B, max_new_tokens=cap, max_latent_candidates=1)
text, (auto, _) = bench_prompt()
tt = F.pad(text.int()[None], (0, 1)).cuda()
nwg = 16 * (B // 4)
for NT in (100, 190):
    ar.prefill(auto.cuda(), tt)
    ar.generate(B, NT, seed=1)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (nwg * 10))()
    assert lib.ttx_attn_stamps(buf, nwg) == 0
    st = np.array(buf, dtype=np.float64).reshape(nwg, 10)
    xcc = st[:, 9].astype(int) & 0xF
    t = (st[:, :9] - st[:, 0].min()) * 0.01
    d = np.diff(t, axis=1)
    ex = t[:, 8]
    mx = np.array([ex[xcc == x].max() for x in range(8)])
    mb = (B * NT + 59) * 16 * 64 * 2 * 2 / 1e6
    print("in situ (capacity %d): last decode-attention launch of a %d-token generation (%.1f MB -> %.1f us at 6.4 TB/s)" % (cap, NT, mb, mb / 6.4))
    print("  workgroup entry p50 %.2f max %.2f | exit p0 %.2f p50 %.2f p90 %.2f max %.2f us | per XCD last exit: %s | harmonic mean %.2f" % (
        np.median(t[:, 0]), t[:, 0].max(), ex.min(), np.median(ex), np.percentile(ex, 90), ex.max(), " ".join("%.1f" % v for v in mx), 8.0 / (1.0 / mx).sum()))
    print("  exit histogram (2 us bins from 0): %s" % np.histogram(ex, bins=np.arange(0, ex.max() + 2, 2))[0].tolist())
    for i, name in enumerate(PH):
        print("  phase %-42s mean %6.2f  p90 %6.2f  max %6.2f us" % (name, d[:, i].mean(), np.percentile(d[:, i], 90), d[:, i].max()))
