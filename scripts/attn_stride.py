#!/usr/bin/env python
"""Does the KV cache's per-(sequence, head) stride (= capacity tmax x 128 B) matter for decode attention?  Last workgroup exit and the spread
between XCDs (phase stamps, -DTT_ATTN_STAMPS build) for several capacities at the same number of own keys.
    TORTOISE_MI355X_LIB=tortoise_tts_amd/lib/libtortoise_mi355x_astamps.so python scripts/attn_stride.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tortoise_tts_amd import engine as E  # noqa: E402

lib = E.init()
lib.ttx_attn_stamps.restype = C.c_int
lib.ttx_attn_stamps.argtypes = [C.c_void_p, C.c_int]
B, H, P1 = 256, 16, 59
dt, tdt = E.TT_F16, torch.float16
g = torch.Generator().manual_seed(0)
q = (torch.randn(B, H * 64, generator=g) * 0.25).to(tdt).cuda()
kp = (torch.randn(H, P1, 64, generator=g) * 2).to(tdt).cuda()
vp = torch.randn(H, P1, 64, generator=g).to(tdt).cuda()
out = torch.zeros(B, H * 64, device="cuda", dtype=tdt)
nwg = H * (B // 4)
VARIANT = int(os.environ.get("ATTN_VARIANT", "0"))
for tgen in (100, 190):
    for tmax in [int(x) for x in os.environ.get("TMAX", "200,201,202,204,205,208,209,216,232,256,264,508,509").split(",")]:
        kc = (torch.randn(B, H, 8, tmax, 8, generator=g) * 2).to(tdt).cuda()
        vc = torch.randn(B, H, tmax, 64, generator=g).to(tdt).cuda()
        res = []
        for rep in range(3):
            for _ in range(3):
                E.check(lib.tt_op_decode_attention(dt, E.ptr(q), E.ptr(kp), E.ptr(vp), P1, E.ptr(kc), E.ptr(vc), tmax, tgen, E.ptr(out), B, H, VARIANT, None))
            torch.cuda.synchronize()
            buf = (C.c_ulonglong * (nwg * 10))()
            assert lib.ttx_attn_stamps(buf, nwg) == 0
            st = np.array(buf, dtype=np.float64).reshape(nwg, 10)
            xcc = st[:, 9].astype(int) & 0xF
            ex = (st[:, 8] - st[:, 0].min()) * 0.01
            mx = np.array([ex[xcc == x].max() for x in range(8)])
            res.append((ex.max(), mx.min(), 8.0 / (1.0 / mx).sum()))
        r = np.array(res)
        print("stride: %3d own keys, capacity %3d (%6d B per head, %7.2f pages of 4 KB per sequence): last exit %5.2f us (%s), fastest XCD %5.2f, harmonic mean of the XCDs %5.2f" % (
            tgen, tmax, tmax * 128, 16 * tmax * 128 / 4096.0, r[:, 0].mean(), " ".join("%.1f" % v for v in r[:, 0]), r[:, 1].mean(), r[:, 2].mean()))
        del kc, vc
