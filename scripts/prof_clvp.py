#!/usr/bin/env python
"""rocprofv3-free per-kernel view of the CLVP stage at the benchmark shape: wall time of score() + the engine's per-class dispatch timings."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from tortoise_tts_amd import engine as E, stages, weights as W  # noqa: E402
from tortoise_tts_amd.config import CLVPConfig  # noqa: E402

lib = E.init()
cfg = CLVPConfig()
sd = W.synthetic_state_dict(W.clvp_manifest(cfg), 1235)
st = stages.ClvpStage(sd, cfg, max_rows=256 * 200)
g = torch.Generator().manual_seed(1)
text = torch.randint(1, 255, (1, 55), generator=g).cuda()
codes = torch.randint(0, 8192, (256, 200), generator=g).cuda()
for _ in range(2):
    st.score(text, codes)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    st.score(text, codes)
torch.cuda.synchronize()
print("clvp score(256 x 200): %.2f ms" % (1e3 * (time.perf_counter() - t0) / 5))
if "--trace" not in sys.argv:
    lib.tt_prof_enable(1)
    st.score(text, codes)
    torch.cuda.synchronize()
    lib.tt_prof_enable(0)
    buf = (C.c_double * 4)()
    for i in range(lib.tt_prof_classes()):
        lib.tt_prof_read(i, buf)
        if buf[0] > 0:
            print("prof %-34s %5d launches %9.2f us avg %8.2f ms total  %7.1f TFLOP/s" % (lib.tt_prof_class_name(i).decode(), int(buf[0]), 1e3 * buf[1] / buf[0], buf[1], buf[2] / max(buf[1], 1e-9) / 1e9))
