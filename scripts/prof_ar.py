#!/usr/bin/env python
"""Per-kernel-class dispatch timings of a short generation at the benchmark shape (eager launches, tt_prof_*)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from bench import bench_prompt  # noqa: E402
from tortoise_tts_amd import engine as E, stages, weights as W  # noqa: E402
from tortoise_tts_amd.config import ARConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--candidates", type=int, default=256)
ap.add_argument("--tokens", type=int, default=40)
ap.add_argument("--fused", type=int, default=0)
ap.add_argument("--capacity", type=int, default=0, help="KV slots per sequence (default: tokens + 8)")
args = ap.parse_args()
B, NT = args.candidates, args.tokens
lib = E.init()
cfg = ARConfig()
sd = W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(cfg), 1234), cfg)
ar = stages.ArStage(sd, cfg, max_batch=B, max_new_tokens=args.capacity or max(64, NT + 8), max_latent_candidates=1)
text, (auto, _) = bench_prompt()
tt = F.pad(text.int()[None], (0, 1)).cuda()
ar.prefill(auto.cuda(), tt)
codes = ar.generate(B, NT, seed=1)[0]
lib.tt_graph_replay(0)
lib.tt_prof_enable(1)
ar.prefill(auto.cuda(), tt)
codes2 = ar.generate(B, NT, seed=1)[0]
torch.cuda.synchronize()
lib.tt_prof_enable(0)
lib.tt_graph_replay(1)
assert torch.equal(codes, codes2)
print("prof AR decode, %d candidates x %d tokens (eager launches, dispatch timestamps):" % (B, NT))
buf = (C.c_double * 4)()
for i in range(lib.tt_prof_classes()):
    lib.tt_prof_read(i, buf)
    if buf[0] > 0:
        print("prof %-34s %6d launches  %7.2f us" % (lib.tt_prof_class_name(i).decode(), int(buf[0]), 1e3 * buf[1] / buf[0]))
