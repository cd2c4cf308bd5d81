#!/usr/bin/env python
"""Where decode attention's time goes: phase stamps (100 MHz wall clock in scalar registers, wave 0 of EVERY workgroup) of
csrc/attention.hip decode_attn_lds_kernel built with -DTT_ATTN_STAMPS, at the benchmark's shape (256 sequences x 16 heads, 59 shared keys).
    python -m tortoise_tts_amd.build --variant astamps -DTT_ATTN_STAMPS
    TORTOISE_MI355X_LIB=tortoise_tts_amd/lib/libtortoise_mi355x_astamps.so python scripts/attn_phases.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tortoise_tts_amd import engine as E  # noqa: E402

PH = ["entry -> prefix staging + first two own-key slots requested", "wait: staged prefix landed, barrier, query", "prefix scores (LDS)",
      "own scores (waits for the K stream)", "first V rows requested, softmax", "prefix PV (LDS)", "own PV (waits for the V stream)", "reduce + store"]
lib = E.init()
lib.ttx_attn_stamps.restype = C.c_int
lib.ttx_attn_stamps.argtypes = [C.c_void_p, C.c_int]
B, H, P1, tmax = 256, 16, 59, 508
dt, tdt = E.TT_F16, torch.float16
g = torch.Generator().manual_seed(0)
q = (torch.randn(B, H * 64, generator=g) * 0.25).to(tdt).cuda()
kp = (torch.randn(H, P1, 64, generator=g) * 2).to(tdt).cuda()
vp = torch.randn(H, P1, 64, generator=g).to(tdt).cuda()
kc = (torch.randn(B, H, 8, tmax, 8, generator=g) * 2).to(tdt).cuda()
vc = torch.randn(B, H, tmax, 64, generator=g).to(tdt).cuda()
out = torch.zeros(B, H * 64, device="cuda", dtype=tdt)
nwg = H * (B // 4)
VARIANT = 0  # the shape's default: decode_attn_lds_kernel, 4 sequences per workgroup
for tgen in (100, 200):
    for _ in range(4):
        E.check(lib.tt_op_decode_attention(dt, E.ptr(q), E.ptr(kp), E.ptr(vp), P1, E.ptr(kc), E.ptr(vc), tmax, tgen, E.ptr(out), B, H, VARIANT, None))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (nwg * 10))()
    assert lib.ttx_attn_stamps(buf, nwg) == 0
    st = np.array(buf, dtype=np.float64).reshape(nwg, 10)
    xcc = st[:, 9].astype(int) & 0xF
    t = (st[:, :9] - st[:, 0].min()) * 0.01  # us since the first workgroup's entry
    d = np.diff(t, axis=1)
    mb = (B * tgen + P1) * H * 64 * 2 * 2 / 1e6
    print("decode attention phases, %d own keys (%.1f MB algorithmic -> %.1f us at 6.4 TB/s): wave 0 of %d workgroups" % (tgen, mb, mb / 6.4, nwg))
    print("  workgroup entry   : p0 %.2f  p50 %.2f  p90 %.2f  max %.2f us   (XCDs seen: %s)" % (t[:, 0].min(), np.median(t[:, 0]), np.percentile(t[:, 0], 90), t[:, 0].max(), sorted(set(xcc.tolist()))))
    print("  workgroup exit    : p0 %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (t[:, 8].min(), np.median(t[:, 8]), np.percentile(t[:, 8], 90), t[:, 8].max()))
    print("  time in workgroup : p50 %.2f  max %.2f us" % (np.median(t[:, 8] - t[:, 0]), (t[:, 8] - t[:, 0]).max()))
    ex = t[:, 8]
    print("  exit by XCD (p50 / max): " + "  ".join("%d: %.1f / %.1f" % (x, np.median(ex[xcc == x]), ex[xcc == x].max()) for x in sorted(set(xcc.tolist()))))
    wgi = np.arange(nwg)
    print("  exit vs workgroup index: corr %.2f; first quarter p50 %.1f, last quarter p50 %.1f; exit histogram (2 us bins from 0): %s" % (
        np.corrcoef(wgi, ex)[0, 1], np.median(ex[: nwg // 4]), np.median(ex[-nwg // 4:]), np.histogram(ex, bins=np.arange(0, ex.max() + 2, 2))[0].tolist()))
    for i, name in enumerate(PH):
        print("  phase %-66s mean %6.2f  p90 %6.2f  max %6.2f us" % (name, d[:, i].mean(), np.percentile(d[:, i], 90), d[:, i].max()))
