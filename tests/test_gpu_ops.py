"""-m gpu: single HIP kernels (through the C-ABI operator entry points) against plain fp32 PyTorch
references computed from the SAME operand-rounded inputs.  Tolerances: relative L2 of the output."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from tortoise_tts_amd import engine as E
from tests.gpu_util import DTYPES, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return E.init()


def dev(t):
    return t.cuda().contiguous()


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("M,N,K", [(200, 136, 128), (70, 130, 64), (2048, 2048, 256), (2048, 1024, 192), (1, 8194, 128)])
def test_gemm_plain(lib, name, dt, tdt, tol, M, N, K):
    g = torch.Generator().manual_seed(M + N)
    A = dev(torch.randn(M, K, generator=g).to(tdt))
    W = dev((torch.randn(N, K, generator=g) / math.sqrt(K)).to(tdt))
    bias = dev(torch.randn(N, generator=g))
    res = dev(torch.randn(M, N, generator=g))
    out = torch.zeros(M, N, device="cuda")
    out_t = torch.zeros(M, N, device="cuda", dtype=tdt)
    E.check(lib.tt_op_gemm(dt, E.ptr(A), K, E.ptr(W), K, M, N, K, 1, 0, 1, E.ptr(bias), E.ACT_GELU_TANH, E.ptr(res), E.ptr(out),
                           E.ptr(out_t), None))
    torch.cuda.synchronize()
    ref = F.gelu(A.float() @ W.float().t() + bias, approximate="tanh") + res
    report(f"gemm {name} {M}x{N}x{K}", out, ref, 2e-5)
    report(f"gemm {name} {M}x{N}x{K} (T out)", out_t.float(), ref, tol)


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("taps", [3, 5])
def test_gemm_conv_taps(lib, name, dt, tdt, tol, taps):
    g = torch.Generator().manual_seed(taps)
    B, S, Cin, Cout = 2, 50, 64, 200
    x = torch.randn(B, S, Cin, generator=g).to(tdt)
    w = (torch.randn(Cout, Cin, taps, generator=g) / math.sqrt(Cin * taps)).to(tdt)
    bias = torch.randn(Cout, generator=g)
    wp = w.permute(0, 2, 1).reshape(Cout, taps * Cin).contiguous()  # [out][tap][in]
    out = torch.zeros(B * S, Cout, device="cuda")
    xd, wd, bd = dev(x), dev(wp), dev(bias)  # keep the device copies alive across the asynchronous launch
    E.check(lib.tt_op_gemm(dt, E.ptr(xd), Cin, E.ptr(wd), taps * Cin, B * S, Cout, taps * Cin, taps, S, 1, E.ptr(bd),
                           E.ACT_NONE, None, E.ptr(out), None, None))
    torch.cuda.synchronize()
    ref = F.conv1d(x.float().permute(0, 2, 1), w.float(), bias, padding=taps // 2).permute(0, 2, 1).reshape(B * S, Cout)
    report(f"conv-gemm {name} taps={taps}", out, ref, 2e-5)


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
def test_gemm_large_m_tile(lib, name, dt, tdt, tol):
    """Shapes with >= 256 tiles of 256 x 256 (gemm.hip pick_tile): the 16-wave tile with the skip read in the epilogue.  Ragged M
    (not a multiple of 16), N = 768 (three column tiles), every output form (run-time epilogue with GELU + skip + both outputs,
    bias + skip -> f32, 3-tap convolution over 20 sequences)."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 22003, 768, 192
    A = dev(torch.randn(M, K, generator=g).to(tdt))
    W = dev((torch.randn(N, K, generator=g) / math.sqrt(K)).to(tdt))
    bias = dev(torch.randn(N, generator=g))
    res = dev(torch.randn(M, N, generator=g))
    out = torch.zeros(M, N, device="cuda")
    out_t = torch.zeros(M, N, device="cuda", dtype=tdt)
    E.check(lib.tt_op_gemm(dt, E.ptr(A), K, E.ptr(W), K, M, N, K, 1, 0, 1, E.ptr(bias), E.ACT_GELU_TANH, E.ptr(res), E.ptr(out), E.ptr(out_t), None))
    torch.cuda.synchronize()
    ref = F.gelu(A.float() @ W.float().t() + bias, approximate="tanh") + res
    report(f"gemm 256-tile {name} gelu + skip {M}x{N}x{K}", out, ref, 2e-5)
    report(f"gemm 256-tile {name} gelu + skip (T out)", out_t.float(), ref, tol)
    out2 = res.clone()                                                                # in place on the skip operand, as the engines call it
    E.check(lib.tt_op_gemm(dt, E.ptr(A), K, E.ptr(W), K, M, N, K, 1, 0, 1, E.ptr(bias), E.ACT_NONE, E.ptr(out2), E.ptr(out2), None, None))
    torch.cuda.synchronize()
    report(f"gemm 256-tile {name} bias + skip in place", out2, A.float() @ W.float().t() + bias + res, 2e-5)
    B, S, Cin, Cout, taps = 20, 870, 64, 1024, 3
    x = torch.randn(B, S, Cin, generator=g).to(tdt)
    w = (torch.randn(Cout, Cin, taps, generator=g) / math.sqrt(Cin * taps)).to(tdt)
    cb = torch.randn(Cout, generator=g)
    xd, wd, bd = dev(x), dev(w.permute(0, 2, 1).reshape(Cout, taps * Cin).contiguous()), dev(cb)
    oc = torch.zeros(B * S, Cout, device="cuda")
    E.check(lib.tt_op_gemm(dt, E.ptr(xd), Cin, E.ptr(wd), taps * Cin, B * S, Cout, taps * Cin, taps, S, 1, E.ptr(bd), E.ACT_NONE, None, E.ptr(oc), None, None))
    torch.cuda.synchronize()
    refc = F.conv1d(xd.float().permute(0, 2, 1), w.float().cuda(), bd, padding=1).permute(0, 2, 1).reshape(B * S, Cout)
    report(f"conv-gemm 256-tile {name} taps=3 over {B} sequences", oc, refc, 2e-5)


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES[:1])
def test_gemm_splitk_slabs(lib, name, dt, tdt, tol):
    g = torch.Generator().manual_seed(7)
    M, N, K, SK = 48, 256, 1024, 4
    A = dev(torch.randn(M, K, generator=g).to(tdt))
    W = dev((torch.randn(N, K, generator=g) / math.sqrt(K)).to(tdt))
    slabs = torch.zeros(SK, M, N, device="cuda")
    E.check(lib.tt_op_gemm(dt, E.ptr(A), K, E.ptr(W), K, M, N, K, 1, 0, SK, None, E.ACT_NONE, None, E.ptr(slabs), None, None))
    torch.cuda.synchronize()
    report("gemm split-K slabs", slabs.sum(0), A.float() @ W.float().t(), 2e-5)


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
def test_row_norms(lib, name, dt, tdt, tol):
    g = torch.Generator().manual_seed(3)
    for D in (128, 768, 1024):
        M = 37
        x = dev(torch.randn(M, D, generator=g) * 3 + 1)
        gam, bet = dev(torch.randn(D, generator=g)), dev(torch.randn(D, generator=g))
        out = torch.zeros(M, D, device="cuda")
        out_t = torch.zeros(M, D, device="cuda", dtype=tdt)
        E.check(lib.tt_op_layernorm(dt, E.ptr(x), M, D, E.ptr(gam), E.ptr(bet), 1e-5, 0, E.ptr(out_t), E.ptr(out), None))
        torch.cuda.synchronize()
        report(f"layernorm D={D}", out, F.layer_norm(x, (D,), gam, bet, 1e-5), 1e-5)
        report(f"layernorm D={D} (T)", out_t.float(), F.layer_norm(x, (D,), gam, bet, 1e-5), tol)
        E.check(lib.tt_op_layernorm(dt, E.ptr(x), M, D, E.ptr(gam), None, 1e-8, 1, None, E.ptr(out), None))
        torch.cuda.synchronize()
        nrm = torch.norm(x, dim=-1, keepdim=True) * D ** -0.5
        report(f"rmsnorm D={D}", out, x / nrm.clamp(min=1e-8) * gam, 1e-5)


@pytest.mark.parametrize("C_", [128, 1024])
def test_groupnorm(lib, C_):
    g = torch.Generator().manual_seed(C_)
    B, S = 2, 77
    x = dev(torch.randn(B, S, C_, generator=g) * 2 + 0.5)
    gam, bet = dev(torch.randn(C_, generator=g)), dev(torch.randn(C_, generator=g))
    ss = dev(torch.randn(B, 2 * C_, generator=g) * 0.3)
    ws = torch.zeros(lib.tt_op_groupnorm_workspace(B, S) // 4 + 16, device="cuda")
    out = torch.zeros(B, S, C_, device="cuda")
    E.check(lib.tt_op_groupnorm(E.TT_BF16, E.ptr(x), B, S, C_, E.ptr(gam), E.ptr(bet), E.ptr(ss), E.ACT_SILU, None, E.ptr(out),
                                E.ptr(ws), None))
    torch.cuda.synchronize()
    y = F.group_norm(x.permute(0, 2, 1), 32, gam, bet, 1e-5)
    y = y * (1 + ss[:, :C_, None]) + ss[:, C_:, None]
    report(f"groupnorm C={C_}", out, F.silu(y).permute(0, 2, 1), 2e-5)


def _attn_ref(q, k, v, causal, relpos):
    # q pre-scaled; q,k,v [B,H,n,64] fp32
    w = q @ k.transpose(-1, -2)
    n = q.shape[2]
    if relpos is not None:
        pos = torch.arange(n, device=q.device)
        d = (pos[None, :] - pos[:, None]).clamp(-64, 64) + 64
        w = w + relpos[:, d][None]
    if causal:
        w = w.masked_fill(torch.triu(torch.ones(n, n, dtype=torch.bool, device=q.device), 1), float("-inf"))
    return torch.softmax(w, dim=-1) @ v


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("mode", ["plain", "causal", "relpos"])
@pytest.mark.parametrize("n,B", [(70, 2), (300, 40), (870, 2), (200, 96), (129, 3), (300, 60)])
def test_flash_attention(lib, name, dt, tdt, tol, mode, n, B):
    """n <= 128 and every causal case: the 16-query-wave kernels (register-prefetch / LDS-staged).  Non-causal n > 128 (round 5):
    flash32_kernel, 32-query waves on v_mfma_f32_32x32x16 - 8 waves with a 2-way key split for launches of < 512 workgroups (n = 870 x 4
    pairs: the denoiser's form; ragged tails 129 / 300; 300 x 120 pairs), 4 waves for more (200 x 1152 pairs)."""
    g = torch.Generator().manual_seed(n)
    H = 2 if n != 200 else 12
    n_pad = (n + 31) // 32 * 32
    q = dev((torch.randn(B, H, n, 64, generator=g) * 0.125 * 2).to(tdt))
    k = dev((torch.randn(B, H, n, 64, generator=g) * 2).to(tdt))
    v = dev(torch.randn(B, H, n, 64, generator=g).to(tdt))
    vt = torch.zeros(B, H, 64, n_pad, device="cuda", dtype=tdt)
    vt[..., :n] = v.transpose(-1, -2)
    relpos = dev(torch.randn(H, 129, generator=g)) if mode == "relpos" else None
    out = torch.zeros(B, n, H * 64, device="cuda", dtype=tdt)
    E.check(lib.tt_op_flash_attention(dt, E.ptr(q), E.ptr(k), E.ptr(vt), E.ptr(out), B, H, n, n_pad, int(mode == "causal"),
                                      E.ptr(relpos), None))
    torch.cuda.synchronize()
    ref = _attn_ref(q.float(), k.float(), v.float(), mode == "causal", relpos).permute(0, 2, 1, 3).reshape(B, n, H * 64)
    report(f"flash {name} {mode} n={n} B={B}", out.float(), ref, tol)


def test_sampler_matches_oracle(lib):
    from oracle import tortoise_oracle as O
    g = torch.Generator().manual_seed(5)
    B, V = 6, 8194
    logits = torch.randn(B, V, generator=g) * 3
    ids = torch.randint(0, V, (B, 20), generator=g)
    ids[:, 0] = 1
    ids[:, 1] = 8192
    q = torch.empty(1, B, V).exponential_(1, generator=g)
    scores = O.warp_logits(logits, ids, 2.0, 0.8, 50, 0.8)
    want = O.multinomial_from_exponential(torch.softmax(scores, -1), q[0])
    import numpy as np
    seen_np = np.zeros((B, (V + 31) // 32), dtype=np.uint32)
    for b in range(B):
        for t in ids[b].tolist():
            seen_np[b, t >> 5] |= np.uint32(1 << (t & 31))
    seen0 = dev(torch.from_numpy(seen_np.view(np.int32)))
    seen = seen0.clone()  # the kernel marks the sampled token as seen
    s = E.Sampling()
    s.temperature, s.top_p, s.repetition_penalty, s.top_k, s.seed, s.row_offset = 0.8, 0.8, 2.0, 50, 0, 0
    qd = dev(q)
    s.exp_noise = E.ptr(qd)
    unfinished = dev(torch.ones(B, dtype=torch.int32))
    codes = torch.zeros(B, 4, device="cuda", dtype=torch.int32)
    logits_d = dev(logits)
    E.check(lib.tt_op_sample(E.ptr(logits_d), V, B, V, E.ptr(seen), C.byref(s), 0, E.ptr(unfinished), 8193, E.ptr(codes), 4, None))
    got = codes[:, 0].cpu().long()
    print("[parity] sampler tokens", got.tolist(), want.tolist())
    assert torch.equal(got, want)
    # Philox path: deterministic, inside the nucleus
    s.exp_noise = None
    s.seed = 1234
    c1 = torch.zeros(B, 4, device="cuda", dtype=torch.int32)
    c2 = torch.zeros(B, 4, device="cuda", dtype=torch.int32)
    for c in (c1, c2):
        un = dev(torch.ones(B, dtype=torch.int32))
        sn = seen0.clone()
        E.check(lib.tt_op_sample(E.ptr(logits_d), V, B, V, E.ptr(sn), C.byref(s), 0, E.ptr(un), 8193, E.ptr(c), 4, None))
    assert torch.equal(c1, c2)
    allowed = torch.isfinite(scores)
    for b in range(B):
        assert allowed[b, int(c1[b, 0])], "sampled token outside the top-k/top-p nucleus"


def test_sampler_tie_plateau_is_deterministic(lib):
    """More scores tie at the k-th value than the survivor buffer holds (quantised / saturated logits): the kept set must not depend
    on atomic slot order.  Rows: a full plateau, and 20 clear winners above a plateau."""
    B, V = 2, 8194
    logits = torch.zeros(B, V)
    logits[1, 100:120] = 5.0
    g = torch.Generator().manual_seed(6)
    q = torch.empty(1, B, V).exponential_(1, generator=g)
    seen_np = torch.zeros(B, (V + 31) // 32, dtype=torch.int32)
    s = E.Sampling()
    s.temperature, s.top_p, s.repetition_penalty, s.top_k, s.seed, s.row_offset = 1.0, 1.0, 1.0, 50, 0, 0
    qd = dev(q)
    s.exp_noise = E.ptr(qd)
    logits_d = dev(logits)
    outs = []
    for _ in range(4):
        codes = torch.zeros(B, 4, device="cuda", dtype=torch.int32)
        un = dev(torch.ones(B, dtype=torch.int32))
        sn = dev(seen_np.clone())
        E.check(lib.tt_op_sample(E.ptr(logits_d), V, B, V, E.ptr(sn), C.byref(s), 0, E.ptr(un), 8193, E.ptr(codes), 4, None))
        outs.append(codes[:, 0].cpu())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    # the deterministic kept set: everything above the plateau + the lowest-index ties up to 512 entries; uniform p -> argmax(1/q) = argmin q
    want0 = int(q[0, 0, :512].argmin())
    kept1 = torch.cat([torch.arange(100, 120), torch.arange(0, 512 - 20)])
    p1 = torch.softmax(logits[1, kept1], -1)
    want1 = int(kept1[(p1 / q[0, 1, kept1]).argmax()])
    print("[parity] sampler plateau tokens", outs[0].tolist(), [want0, want1])
    assert outs[0].tolist() == [want0, want1]


def test_univnet_kernels(lib):
    g = torch.Generator().manual_seed(9)
    # dilated conv with fused LeakyReLUs
    T = 700
    x = dev(torch.randn(32, T, generator=g))
    w = dev(torch.randn(32, 32, 3, generator=g) * 0.1)
    b = dev(torch.randn(32, generator=g) * 0.1)
    for dil in (1, 3, 9, 27):
        y = torch.zeros(32, T, device="cuda")
        E.check(lib.tt_op_conv1d(E.ptr(x), E.ptr(w), E.ptr(b), E.ptr(y), 32, 32, T, 3, dil, 0, 0.2, E.ACT_LRELU, 0.2, None))
        torch.cuda.synchronize()
        ref = F.leaky_relu(F.conv1d(F.leaky_relu(x, 0.2)[None], w, b, padding=dil, dilation=dil), 0.2)[0]
        report(f"conv1d dil={dil}", y, ref, 1e-5)
    # reflect-padded k7 convs (conv_pre 64->32, conv_post 32->1 + tanh)
    z = dev(torch.randn(64, 90, generator=g))
    w7 = dev(torch.randn(32, 64, 7, generator=g) * 0.05)
    y = torch.zeros(32, 90, device="cuda")
    E.check(lib.tt_op_conv1d(E.ptr(z), E.ptr(w7), E.ptr(b), E.ptr(y), 64, 32, 90, 7, 1, 1, -1.0, E.ACT_NONE, 0.0, None))
    torch.cuda.synchronize()
    report("conv_pre", y, F.conv1d(F.pad(z[None], (3, 3), mode="reflect"), w7, b)[0], 1e-5)
    wp = dev(torch.randn(1, 32, 7, generator=g) * 0.1)
    bp = dev(torch.randn(1, generator=g))
    y1 = torch.zeros(1, T, device="cuda")
    E.check(lib.tt_op_conv1d(E.ptr(x), E.ptr(wp), E.ptr(bp), E.ptr(y1), 32, 1, T, 7, 1, 1, 0.2, E.ACT_TANH, 0.0, None))
    torch.cuda.synchronize()
    report("conv_post", y1, torch.tanh(F.conv1d(F.pad(F.leaky_relu(x, 0.2)[None], (3, 3), mode="reflect"), wp, bp))[0], 1e-5)
    # transposed conv
    for stride in (8, 4):
        Tin = 100
        xi = dev(torch.randn(32, Tin, generator=g))
        wt = dev(torch.randn(32, 32, 2 * stride, generator=g) * 0.1)
        yt = torch.zeros(32, Tin * stride, device="cuda")
        E.check(lib.tt_op_convt1d(E.ptr(xi), E.ptr(wt), E.ptr(b), E.ptr(yt), 32, Tin, stride, 0.2, None))
        torch.cuda.synchronize()
        ref = F.conv_transpose1d(F.leaky_relu(xi, 0.2)[None], wt, b, stride=stride, padding=stride // 2 + stride % 2,
                                 output_padding=stride % 2)[0]
        report(f"convt stride={stride}", yt, ref, 1e-5)
    # LVC + gate
    from oracle import tortoise_oracle as O
    for hop in (8, 64, 256):
        L = 7
        T = L * hop
        xin = dev(torch.randn(32, T, generator=g))
        xres = dev(torch.randn(32, T, generator=g))
        kern = dev(torch.randn(L, 4 * 6144, generator=g) * 0.1)
        kb = dev(torch.randn(L, 256, generator=g) * 0.1)
        j = 2
        xr = xres.clone()
        # round 6: the predicted kernels arrive in the operand type the KernelPredictor GEMM wrote them in (f32 in the verification mode);
        # the arithmetic is f32 on the widened values either way, so every form must match the oracle on the SAME (rounded) kernels
        for kdt, tdt_k, kname in ((E.TT_F32, torch.float32, "f32"), (E.TT_F16, torch.float16, "f16"), (E.TT_BF16, torch.bfloat16, "bf16")):
            kern_t = kern.to(tdt_k).contiguous()
            xr = xres.clone()
            E.check(lib.tt_op_lvc(kdt, E.ptr(xin), E.ptr(kern_t), 4 * 6144, j * 6144, E.ptr(kb), 256, j * 64, E.ptr(xr), L, hop, None))
            torch.cuda.synchronize()
            kk = kern_t.float()[:, j * 6144:(j + 1) * 6144].reshape(L, 32, 64, 3).permute(1, 2, 3, 0)[None].cpu()  # [1, i, o, k, L]
            bb = kb[:, j * 64:(j + 1) * 64].t()[None].cpu()
            o = O.location_variable_convolution(xin[None].cpu(), kk, bb, hop)
            ref = xres.cpu() + (torch.sigmoid(o[:, :32]) * torch.tanh(o[:, 32:]))[0]
            report(f"lvc hop={hop} kernels {kname}", xr, ref, 1e-5)


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("B,S,N", [(2, 870, 1024), (1, 870, 1024), (3, 333, 256), (8, 33, 512), (1, 4096, 1024), (2, 129, 1024)])
def test_groupnorm_silu_on_the_gemm_a_path(lib, name, dt, tdt, tol, B, S, N):
    """The fused ResBlock in_layers launch (csrc/gemm_gna.h, diffusion_decoder.py:60-80 GroupNorm32 -> SiLU -> 1x1 conv) against torch
    fp32: samples whose boundary falls inside a 32-row tile (S = 870, 333, 33, 129), a row count that is no multiple of the tile
    (2 x 129, 3 x 333), one sample, the largest row count the kernel takes; activations with a per-group offset and spread so that the
    mean / rstd folding has something to cancel.  The reference rounds the normalised activations to the operand type as the kernel does."""
    if dt == E.TT_F32:
        pytest.skip("the fused launch is a 16-bit-operand kernel (the fp32 verification mode keeps the stand-alone apply)")
    g = torch.Generator().manual_seed(B * 1000 + S)
    C = 1024
    x = torch.randn(B, S, C, generator=g) * (0.5 + torch.rand(1, 1, C, generator=g) * 3.0) + torch.randn(1, 1, C, generator=g) * 2.0
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    W = (torch.randn(N, C, generator=g) / math.sqrt(C)).to(tdt)
    bias = torch.randn(N, generator=g)
    xd, gd, bd, Wd, biasd = dev(x), dev(gamma), dev(beta), dev(W), dev(bias)
    out = torch.zeros(B * S, N, device="cuda")
    ws = torch.zeros(lib.tt_op_gn_gemm_workspace(B, S) // 4, device="cuda")
    E.check(lib.tt_op_gn_gemm(dt, E.ptr(xd), B, S, E.ptr(gd), E.ptr(bd), E.ACT_SILU, E.ptr(Wd), E.ptr(biasd), N, E.ptr(out), E.ptr(ws), None))
    torch.cuda.synchronize()
    h = F.silu(F.group_norm(xd.transpose(1, 2), 32, gd, bd, eps=1e-5)).transpose(1, 2).reshape(B * S, C)
    ref = h.to(tdt).float() @ Wd.float().t() + biasd
    report(f"groupnorm + silu on the GEMM A path {name} B={B} S={S} N={N}", out, ref, 2e-4)
    assert torch.isfinite(out).all()
