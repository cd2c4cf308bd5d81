#!/usr/bin/env python
"""Where the sampling kernel's time goes: phase stamps (100 MHz wall clock, thread 0 of block 0) of csrc/sampling.hip built with
-DTT_SAMPLE_STAMPS, on the logits of a real decode step of the benchmark's model.
    python -m tortoise_tts_amd.build --variant stamps -DTT_SAMPLE_STAMPS
    TORTOISE_MI355X_LIB=tortoise_tts_amd/lib/libtortoise_mi355x_stamps.so python scripts/sample_phases.py"""
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

PHASES = ["state + logits / seen loads, penalty, temperature", "keys, thread maxima -> LDS, bound of the k-th key (ballot counts)", "candidate compaction",
          "all-pairs counts (four waves)", "places: scatter into sorted order", "exp + sequential total", "(divisions: plateau path only)", "top-p walk + kept total", "Philox + p / q", "argmax + commit",
          "next step's embedding row"]
lib = E.init()
lib.ttx_sample_stamps.restype = C.c_int
lib.ttx_sample_stamps.argtypes = [C.c_void_p]
cfg = ARConfig()
sd = W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(cfg), 1234), cfg)
for B in (1, 256):
    ar = stages.ArStage(sd, cfg, max_batch=max(B, 8), max_new_tokens=64, max_latent_candidates=1)
    text, (auto, _) = bench_prompt()
    tt = F.pad(text.int()[None], (0, 1)).cuda()
    ar.prefill(auto.cuda(), tt)
    codes = ar.generate(B, 40, seed=1)[0]
    logits = ar.logits(B).float().contiguous()
    V = logits.shape[1]
    seen = np.zeros((B, (V + 31) // 32), dtype=np.uint32)
    for b in range(B):
        for t in codes[b].tolist():
            seen[b, t >> 5] |= np.uint32(1 << (t & 31))
    seen_d = torch.from_numpy(seen.view(np.int32)).cuda()
    s = E.Sampling()
    s.temperature, s.top_p, s.repetition_penalty, s.top_k, s.seed, s.row_offset = 0.8, 0.8, 2.0, 50, 7, 0
    acc = np.zeros(11)
    reps = 20
    for r in range(reps):
        un = torch.ones(B, dtype=torch.int32, device="cuda")
        out = torch.zeros(B, 4, dtype=torch.int32, device="cuda")
        E.check(lib.tt_op_sample(E.ptr(logits), V, B, V, E.ptr(seen_d.clone()), C.byref(s), 0, E.ptr(un), 8193, E.ptr(out), 4, None))
        st = (C.c_ulonglong * 16)()
        assert lib.ttx_sample_stamps(st) == 0
        acc += np.diff(np.array(st[:12], dtype=np.float64)) * 0.01  # 100 MHz -> us
        ncand = int(st[15])
    acc /= reps
    print("sampler phases, B = %d (block 0, mean of %d launches; %d top-k candidates in the last one):" % (B, reps, ncand))
    for name, us in zip(PHASES, acc):
        print("phase  %-58s %6.2f us" % (name, us))
    print("phase  %-58s %6.2f us" % ("sum (first to last stamp)", acc.sum()))
    ar.close()
