"""-m gpu, round 5: the five-launch decode step (csrc/gpt2.hip; reference work: transformers GPT2Block.forward as
tortoise/models/autoregressive.py:150-163 runs it per token, loop at api.py:415-427).
  * operator level (tt_op_resid_ln): the projection with the in-launch split-K fold (arrival tickets / one-workgroup fold) against fp32
    torch from the same operand-rounded inputs - residual rows, their T copy, the per-band LayerNorm statistics; the following GEMM with
    LayerNorm folded in against (a) the same folded algebra in fp32 and (b) plain F.layer_norm + Linear + gelu_new; the ticket fold is
    bit-identical to the one-workgroup fold and to itself over repetitions (it sums the slabs in slab order whoever arrives last);
  * engine level: five launches per layer, codes deterministic over repetitions, graph == eager, a row's logits do not depend on the
    batch size, the two forms of the step agree to operand rounding, ragged stop tokens leave the loop at the same step."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import make_golden as G
from tortoise_tts_amd import engine as E
from tortoise_tts_amd import stages
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.config import ARConfig
from tests.gpu_util import DTYPES, report, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return E.init()


def _resid_ln(lib, dt, tdt, M, D, K, splitk, N2, seed, x0=None):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g).to(tdt).cuda()
    Wp = (torch.randn(D, K, generator=g) / math.sqrt(K)).to(tdt).cuda()
    bias = torch.randn(D, generator=g).cuda()
    x = (torch.randn(M, D, generator=g) * 2.0 + 0.3).cuda() if x0 is None else x0.clone()
    x_in = x.clone()
    gamma = (1.0 + 0.2 * torch.randn(D, generator=g)).double()
    beta = (0.1 * torch.randn(D, generator=g)).double()
    W2 = (torch.randn(N2, D, generator=g) / math.sqrt(D)).double()
    b2 = torch.randn(N2, generator=g).double()
    Wg = (W2 * gamma[None, :]).float().to(tdt).cuda()
    colsum = Wg.double().sum(dim=1).float().cuda()
    bias2 = (b2 + W2 @ beta).float().cuda()
    xt = torch.zeros(M, D, dtype=tdt, device="cuda")
    stats = torch.zeros(M, D // 32, 2, device="cuda")
    out = torch.zeros(M, N2, dtype=tdt, device="cuda")
    E.check(lib.tt_op_resid_ln(dt, E.ptr(A), K, E.ptr(Wp), E.ptr(bias), E.ptr(x), M, D, splitk, E.ptr(Wg), E.ptr(colsum), E.ptr(bias2), N2,
                               E.ptr(out), E.ptr(xt), E.ptr(stats), None))
    torch.cuda.synchronize()
    return dict(A=A, Wp=Wp, bias=bias, x_in=x_in, x=x, xt=xt, stats=stats, out=out, Wg=Wg, colsum=colsum, bias2=bias2,
                gamma=gamma, beta=beta, W2=W2, b2=b2)


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES)
@pytest.mark.parametrize("M,D,K,splitk", [(256, 1024, 1024, 4), (256, 1024, 4096, 4), (100, 1024, 1024, 4), (16, 1024, 4096, 8), (37, 1024, 1024, 2),
                                           (1300, 1024, 1024, -4), (256, 1024, 4096, -4), (64, 128, 128, 1), (70, 128, 512, 2), (64, 256, 1024, 4)])
def test_projection_with_in_launch_fold_and_folded_layernorm(lib, name, dt, tdt, tol, M, D, K, splitk):
    N2 = 4 * D if D <= 256 else 2048
    r = _resid_ln(lib, dt, tdt, M, D, K, splitk, N2, seed=M + K + splitk)
    tag = f"resid {name} M={M} D={D} K={K} splitk={splitk}"
    ref_x = r["x_in"] + r["bias"] + r["A"].float() @ r["Wp"].float().t()
    report(f"{tag}: x += A W^T + b", r["x"], ref_x, 2e-5)
    assert torch.equal(r["xt"], r["x"].to(tdt)), f"{tag}: the T copy is not the rounded f32 row"
    xb = r["x"].double().reshape(M, D // 32, 32)
    report(f"{tag}: band sums", r["stats"][..., 0], xb.sum(-1).float(), 1e-5)
    report(f"{tag}: band sums of squares", r["stats"][..., 1], (xb * xb).sum(-1).float(), 1e-5)
    # the following GEMM: (a) the folded algebra in fp64 from the operands the kernel saw
    xd = r["x"].double()
    mean = xd.mean(dim=1, keepdim=True)
    rstd = 1.0 / torch.sqrt(xd.var(dim=1, unbiased=False, keepdim=True) + 1e-5)
    pre = rstd * (r["xt"].double() @ r["Wg"].double().t() - mean * r["colsum"].double()[None, :]) + r["bias2"].double()[None, :]
    report(f"{tag}: folded LayerNorm GEMM vs the same algebra in fp64 (T out)", r["out"].float(), F.gelu(pre.float(), approximate="tanh"), 6e-3 if tdt == torch.bfloat16 else 8e-4)
    # (b) what the reference computes: LayerNorm -> Linear -> gelu_new in fp32 with the unfolded fp32 weights
    ln = F.layer_norm(r["x"].double(), (D,), r["gamma"].cuda(), r["beta"].cuda(), 1e-5)
    ref = F.gelu((ln @ r["W2"].cuda().t() + r["b2"].cuda()).float(), approximate="tanh")
    report(f"{tag}: folded LayerNorm GEMM vs layer_norm + Linear + gelu_new fp32", r["out"].float(), ref, tol)


@pytest.mark.parametrize("name,dt,tdt,tol", DTYPES[:1])
def test_ticket_fold_is_order_independent(lib, name, dt, tdt, tol):
    """Which K range arrives last varies from launch to launch; the last arriver sums the slabs in slab order, so the residual rows are
    the same bits every time - and the bits of the one-workgroup fold over the same four ranges."""
    M, D, K = 256, 1024, 4096
    g = torch.Generator().manual_seed(5)
    x0 = (torch.randn(M, D, generator=g) * 3.0).cuda()
    base = _resid_ln(lib, dt, tdt, M, D, K, -4, 2048, seed=99, x0=x0)
    for rep in range(25):
        r = _resid_ln(lib, dt, tdt, M, D, K, 4, 2048, seed=99, x0=x0)
        assert torch.equal(r["x"], base["x"]), f"repetition {rep}: ticket fold differs from the one-workgroup fold in {int((r['x'] != base['x']).sum())} elements"
        assert torch.equal(r["stats"], base["stats"]) and torch.equal(r["out"], base["out"])
    print("[parity] ticket fold == one-workgroup fold, 25 repetitions: bit-identical")


def _stage(cfg, sd, B, new=48):
    return stages.ArStage(sd, cfg, dtype=E.TT_BF16, max_batch=B, max_text=40, max_new_tokens=new, max_latent_candidates=1)


@pytest.mark.parametrize("eos_boost", [None, 3.0])
@torch.no_grad()
def test_five_launch_decode_step_is_deterministic_and_batch_independent(eos_boost):
    cfg = ARConfig(**G.AR_CFG)
    sd = G.sampling_state_dict(cfg, eos_boost) if eos_boost else W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(cfg), seed=G.AR_SEED), cfg)
    cond, text = G.ar_inputs(cfg)
    B, steps = 64, 40
    st = _stage(cfg, sd, B)
    assert st.stat(3) == 0 and st.stat(2) == 7 * cfg.layers + 4, "the seven-launch form is the measured default"
    st.set_option(E.TT_AR_OPT_FUSED_STEP, 1)
    assert st.stat(3) == 1 and st.stat(2) == 5 * cfg.layers + 5, f"decode step: {st.stat(2)} launches, five-launch form {st.stat(3)}"
    st.prefill(cond, text)
    base, n0 = st.generate(B, steps, seed=11)
    base = base.clone()
    if eos_boost:
        stop = cfg.stop_mel_token
        ends = [(int((r == stop).nonzero()[0]) if (r == stop).any() else steps) for r in base.cpu()]
        assert min(ends) < max(ends), "rows did not finish at different steps: the test lost its point"
    for rep in range(12):
        st.prefill(cond, text)
        got, n = st.generate(B, steps, seed=11)
        assert n == n0 and torch.equal(got, base), f"repetition {rep} of the five-launch step changed the codes"
    E.load_library().tt_graph_replay(0)
    try:
        for rep in range(3):
            st.prefill(cond, text)
            got, n = st.generate(B, steps, seed=11)
            assert n == n0 and torch.equal(got, base), f"eager launches differ from the replayed graph (repetition {rep})"
    finally:
        E.load_library().tt_graph_replay(1)
    assert st.stat(1) == 0
    # a row's logits do not depend on the batch it is decoded in, and the two forms of the step agree to operand rounding
    toks = base[:, :3].int()
    lg = {}
    for fused, nb in ((1, B), (1, 16), (0, B)):
        st.set_option(E.TT_AR_OPT_FUSED_STEP, fused)
        assert st.stat(3) == fused
        st.prefill(cond, text)
        st.begin(nb)
        for j in range(3):
            st.decode_step(toks[:nb, j].contiguous())
        lg[(fused, nb)] = st.logits(nb).clone()
    assert torch.equal(lg[(1, B)][:16], lg[(1, 16)]), "logits of a row depend on the batch size in the five-launch form"
    report("decode step: five-launch vs seven-launch logits (bf16 operands, 3 teacher-forced steps)", lg[(1, B)], lg[(0, B)], 2.5e-2)
    st.close()
