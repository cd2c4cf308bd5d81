"""Helpers shared by the -m gpu parity tests."""
import torch

from tortoise_tts_amd import engine as E

DTYPES = [("bf16", E.TT_BF16, torch.bfloat16, 2.5e-2), ("f16", E.TT_F16, torch.float16, 4e-3)]


def quantize_sd(sd, tdtype):
    """Round every matrix / conv kernel to the engine's operand type so the oracle and the engine
    consume bit-identical weights; what remains is activation rounding (stated per test)."""
    out = {}
    for k, v in sd.items():
        out[k] = v.to(tdtype).float() if (v.dim() >= 2 and v.is_floating_point()) else v
    return out


def rel_err(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def max_err(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


def report(name, a, b, tol):
    r, m = rel_err(a, b), max_err(a, b)
    print(f"[parity] {name}: rel_l2={r:.3e} max_abs={m:.3e} (tol rel_l2 {tol:.1e})")
    assert r == r and r < tol, f"{name}: rel_l2 {r:.3e} >= {tol:.1e} (max_abs {m:.3e})"
    return r
