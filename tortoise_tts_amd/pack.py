"""Reference-layout state_dicts -> device-resident engine weights.

Input: the state_dicts the reference loads at tortoise/api.py:221-237 (autoregressive.pth,
clvp2.pth, diffusion_decoder.pth, vocoder.pth['model_g']), fp32 on CPU.  Output: contiguous device
tensors in the layouts include/tortoise_mi355x.h documents (GEMM weights [out][in] in the MFMA
operand type, conv kernels [out][tap][in_padded], norms / biases / embeddings f32) plus the ctypes
structs that point at them.  Every tensor is kept alive by the returned holder object.
"""
import math
import ctypes as C

import torch

from . import engine as E
from .config import ARConfig, CLVPConfig, CVVPConfig, DiffusionConfig, VocoderConfig


def torch_dtype(dtype):
    return {E.TT_BF16: torch.bfloat16, E.TT_F16: torch.float16, E.TT_F32: torch.float32}[dtype]


class Holder:
    """Keeps packed tensors + ctypes arrays alive and offers short helpers."""

    def __init__(self, device, dtype):
        self.device = device
        self.dtype = dtype
        self.tdtype = torch_dtype(dtype)
        self.keep = []

    def f32(self, t):
        t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self.keep.append(t)
        return t

    def op(self, t):
        """GEMM operand: [out][in...] flattened to 2-D, operand dtype."""
        t = t.detach().to(device=self.device, dtype=torch.float32)
        t = t.reshape(t.shape[0], -1).to(self.tdtype).contiguous()
        self.keep.append(t)
        return t

    def conv(self, w, in_pad=None):
        """Conv1d weight [out][in][k] -> [out][k][in_pad] operand."""
        w = w.detach().to(device=self.device, dtype=torch.float32)
        out_c, in_c, k = w.shape
        in_pad = in_pad or in_c
        p = torch.zeros(out_c, k, in_pad, device=self.device, dtype=torch.float32)
        p[:, :, :in_c] = w.permute(0, 2, 1)
        return self.op(p)


def _p(t):
    return t.data_ptr() if t is not None else None


# ----------------------------------------------------------------------------------------- AR
def pack_ar(sd, cfg: ARConfig, device, dtype):
    h = Holder(device, dtype)
    wop = h.op
    layers = (E.GptLayer * cfg.layers)()
    for i in range(cfg.layers):
        p = f"gpt.h.{i}"
        L = layers[i]
        L.ln1_g = _p(h.f32(sd[f"{p}.ln_1.weight"]))
        L.ln1_b = _p(h.f32(sd[f"{p}.ln_1.bias"]))
        L.w_qkv = _p(wop(sd[f"{p}.attn.c_attn.weight"].t()))      # HF Conv1D stores [in, out]
        L.b_qkv = _p(h.f32(sd[f"{p}.attn.c_attn.bias"]))
        L.w_proj = _p(wop(sd[f"{p}.attn.c_proj.weight"].t()))
        L.b_proj = _p(h.f32(sd[f"{p}.attn.c_proj.bias"]))
        L.ln2_g = _p(h.f32(sd[f"{p}.ln_2.weight"]))
        L.ln2_b = _p(h.f32(sd[f"{p}.ln_2.bias"]))
        L.w_fc = _p(wop(sd[f"{p}.mlp.c_fc.weight"].t()))
        L.b_fc = _p(h.f32(sd[f"{p}.mlp.c_fc.bias"]))
        L.w_proj2 = _p(wop(sd[f"{p}.mlp.c_proj.weight"].t()))
        L.b_proj2 = _p(h.f32(sd[f"{p}.mlp.c_proj.bias"]))
    w = E.ArWeights()
    w.layers_host = layers
    w.lnf_g = _p(h.f32(sd["gpt.ln_f.weight"]))
    w.lnf_b = _p(h.f32(sd["gpt.ln_f.bias"]))
    w.final_norm_g = _p(h.f32(sd["final_norm.weight"]))
    w.final_norm_b = _p(h.f32(sd["final_norm.bias"]))
    w.w_mel_head = _p(wop(sd["mel_head.weight"]))
    w.b_mel_head = _p(h.f32(sd["mel_head.bias"]))
    h.mel_emb = h.f32(sd["mel_embedding.weight"])
    h.mel_pos = h.f32(sd["mel_pos_embedding.emb.weight"])
    h.text_emb = h.f32(sd["text_embedding.weight"])
    h.text_pos = h.f32(sd["text_pos_embedding.emb.weight"])
    w.mel_emb = _p(h.mel_emb)
    w.mel_pos = _p(h.mel_pos)
    h.layers = layers
    h.weights = w
    return h


# ----------------------------------------------------------------------------------------- CLVP
def geglu_interleave(inner):
    """Row order of a GEGLU projection for the engine's fused epilogue (csrc/gemm.h EPI_GEGLU): the reference stores [value 0..inner) |
    gate 0..inner) (xtransformers.py:429-437, `x, gate = proj(x).chunk(2)`); the engine wants them interleaved in strips of 16 -
    [value 0..15 | gate 0..15 | value 16..31 | gate 16..31 | ...].  Returns idx with new[r] = old[idx[r]]."""
    assert inner % 16 == 0
    v = torch.arange(inner).reshape(-1, 16)
    return torch.stack([v, v + inner], dim=1).reshape(-1)


def _pack_xenc_layers(h, sd, base, depth, wrap=".wrap"):
    """The x-transformers Encoder sublayers of `base`.attn_layers as tt_clvp_layer[depth] (q / k / v stacked, GEGLU rows interleaved).
    wrap: CLVP's CheckpointedXTransformerEncoder wraps every sublayer (keys `...layers.N.1.wrap.to_q`), CVVP's plain wrapper does not."""
    layers = (E.ClvpLayer * depth)()
    for li in range(depth):
        pa = f"{base}.attn_layers.layers.{2 * li}"
        pf = f"{base}.attn_layers.layers.{2 * li + 1}"
        L = layers[li]
        L.attn_norm_g = _p(h.f32(sd[f"{pa}.0.0.g"]))
        L.w_qkv = _p(h.op(torch.cat([sd[f"{pa}.1{wrap}.to_q.weight"], sd[f"{pa}.1{wrap}.to_k.weight"],
                                     sd[f"{pa}.1{wrap}.to_v.weight"]], dim=0)))
        L.w_out = _p(h.op(sd[f"{pa}.1{wrap}.to_out.weight"]))
        L.b_out = _p(h.f32(sd[f"{pa}.1{wrap}.to_out.bias"]))
        L.ff_norm_g = _p(h.f32(sd[f"{pf}.0.0.g"]))
        gl = geglu_interleave(sd[f"{pf}.1{wrap}.net.0.proj.weight"].shape[0] // 2)
        L.w_ff1 = _p(h.op(sd[f"{pf}.1{wrap}.net.0.proj.weight"][gl]))
        L.b_ff1 = _p(h.f32(sd[f"{pf}.1{wrap}.net.0.proj.bias"][gl]))
        L.w_ff2 = _p(h.op(sd[f"{pf}.1{wrap}.net.3.weight"]))
        L.b_ff2 = _p(h.f32(sd[f"{pf}.1{wrap}.net.3.bias"]))
    return layers


def _pack_clvp_tower(h, sd, cfg: CLVPConfig, tower, emb_key, latent_key):
    base = f"{tower}.transformer"
    layers = _pack_xenc_layers(h, sd, base, cfg.depth)
    t = E.ClvpTower()
    t.layers_host = layers
    t.emb = _p(h.f32(sd[emb_key]))
    t.inv_freq = _p(h.f32(sd[f"{base}.attn_layers.rotary_pos_emb.inv_freq"]))
    t.norm_g = _p(h.f32(sd[f"{base}.norm.weight"]))
    t.norm_b = _p(h.f32(sd[f"{base}.norm.bias"]))
    t.w_latent = _p(h.op(sd[latent_key]))
    h.keep.append(layers)
    return t


def pack_clvp(sd, cfg: CLVPConfig, device, dtype):
    h = Holder(device, dtype)
    h.text = _pack_clvp_tower(h, sd, cfg, "text_transformer", "text_emb.weight", "to_text_latent.weight")
    h.speech = _pack_clvp_tower(h, sd, cfg, "speech_transformer", "speech_emb.weight", "to_speech_latent.weight")
    h.temperature = h.f32(sd["temperature"].reshape(1))
    return h


def pack_cvvp(sd, cfg: CVVPConfig, device, dtype, mel_pad=128):
    """CVVP.state_dict() (cvvp.py:63-98, the api.py:254 instance) for tt_cvvp_create."""
    assert cfg.latent_multiplier == 1, "the reference builds CVVP with latent_multiplier=1 (api.py:254-255)"
    h = Holder(device, dtype)
    D = cfg.model_dim
    w = E.CvvpWeights()
    for tower, dst in (("conditioning_transformer", w.cond), ("speech_transformer", w.speech)):
        base = f"{tower}.transformer"
        layers = _pack_xenc_layers(h, sd, base, cfg.depth, wrap="")
        h.keep.append(layers)
        dst.layers_host = layers
        dst.inv_freq = _p(h.f32(sd[f"{base}.attn_layers.rotary_pos_emb.inv_freq"]))
        dst.norm_g = _p(h.f32(sd[f"{base}.norm.weight"]))
        dst.norm_b = _p(h.f32(sd[f"{base}.norm.bias"]))
        dst.w_pre0 = _p(h.op(sd[f"{tower}.pre_combiner.0.weight"]))
        dst.b_pre0 = _p(h.f32(sd[f"{tower}.pre_combiner.0.bias"]))
        dst.attn = _pack_attn(h, sd, f"{tower}.pre_combiner.1", D, cfg.heads)
        dst.w_pre2 = _p(h.op(sd[f"{tower}.pre_combiner.2.weight"]))
        dst.b_pre2 = _p(h.f32(sd[f"{tower}.pre_combiner.2.bias"]))
    w.cond.w_latent = _p(h.op(sd["to_conditioning_latent.weight"]))
    w.speech.w_latent = _p(h.op(sd["to_speech_latent.weight"]))
    w.w_cond0 = _p(h.conv(sd["cond_emb.0.weight"], mel_pad))
    w.b_cond0 = _p(h.f32(sd["cond_emb.0.bias"]))
    w.w_cond1 = _p(h.conv(sd["cond_emb.1.weight"]))
    w.b_cond1 = _p(h.f32(sd["cond_emb.1.bias"]))
    w.speech_emb = _p(h.f32(sd["speech_emb.emb.weight"]))
    w.temperature = _p(h.f32(sd["temperature"].reshape(1)))
    h.weights = w
    h.mel_pad = mel_pad
    return h


# ----------------------------------------------------------------------------------------- diffusion
def _rel_pos_bucket(rel, num_buckets=32, max_distance=64):
    # RelativePositionBias._relative_position_bucket, causal=False (xtransformers.py:155-175); rel = k - q
    nb = num_buckets // 2
    n = -rel
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def relpos_table(weight, scale=8.0):
    """[heads][129]: additive attention bias for clamp(k - q, -64, 64) (the bucket function saturates
    well before |k - q| = 64, so the clamp is exact).  weight: [32 buckets][heads]."""
    d = torch.arange(-64, 65)
    bucket = _rel_pos_bucket(d)
    return (weight.detach().float().cpu()[bucket] * scale).t().contiguous()  # [heads][129]


def _qkv_head_major_perm(C, heads):
    """Reference AttentionBlock qkv channels are [head][q|k|v][64] (arch_util.py:63); the engine wants
    [q|k|v][head][64].  Returns idx with new[n] = old[idx[n]]."""
    ch = C // heads
    idx = torch.empty(3 * C, dtype=torch.long)
    for part in range(3):
        for hd in range(heads):
            dst = part * C + hd * ch
            src = hd * 3 * ch + part * ch
            idx[dst:dst + ch] = torch.arange(src, src + ch)
    return idx


def _pack_attn(h, sd, prefix, C, heads):
    a = E.AttnBlock()
    perm = _qkv_head_major_perm(C, heads)
    a.norm_g = _p(h.f32(sd[f"{prefix}.norm.weight"]))
    a.norm_b = _p(h.f32(sd[f"{prefix}.norm.bias"]))
    a.w_qkv = _p(h.op(sd[f"{prefix}.qkv.weight"][perm]))
    a.b_qkv = _p(h.f32(sd[f"{prefix}.qkv.bias"][perm]))
    a.w_proj = _p(h.op(sd[f"{prefix}.proj_out.weight"]))
    a.b_proj = _p(h.f32(sd[f"{prefix}.proj_out.bias"]))
    key = f"{prefix}.relative_pos_embeddings.relative_attention_bias.weight"
    a.relpos = _p(h.f32(relpos_table(sd[key], (C // heads) ** 0.5))) if key in sd else None
    return a


def _pack_res(h, sd, prefix):
    r = E.ResBlock()
    r.gn1_g = _p(h.f32(sd[f"{prefix}.in_layers.0.weight"]))
    r.gn1_b = _p(h.f32(sd[f"{prefix}.in_layers.0.bias"]))
    r.w_in = _p(h.op(sd[f"{prefix}.in_layers.2.weight"]))
    r.b_in = _p(h.f32(sd[f"{prefix}.in_layers.2.bias"]))
    r.gn2_g = _p(h.f32(sd[f"{prefix}.out_layers.0.weight"]))
    r.gn2_b = _p(h.f32(sd[f"{prefix}.out_layers.0.bias"]))
    r.w_out = _p(h.conv(sd[f"{prefix}.out_layers.3.weight"]))
    r.b_out = _p(h.f32(sd[f"{prefix}.out_layers.3.bias"]))
    return r


def pack_diffusion(sd, cfg: DiffusionConfig, device, dtype, in_pad=128):
    h = Holder(device, dtype)
    C, H, L = cfg.model_channels, cfg.num_heads, cfg.num_layers
    NR = 3 + L + 3
    res = (E.ResBlock * NR)()
    attn = (E.AttnBlock * (3 + L))()
    lat = (E.AttnBlock * 4)()
    res_prefixes = [f"conditioning_timestep_integrator.{i}.resblk" for i in range(3)] + \
                   [f"layers.{i}.resblk" for i in range(L)] + [f"layers.{i}" for i in range(L, L + 3)]
    attn_prefixes = [f"conditioning_timestep_integrator.{i}.attn" for i in range(3)] + [f"layers.{i}.attn" for i in range(L)]
    for i, p in enumerate(res_prefixes):
        res[i] = _pack_res(h, sd, p)
    for i, p in enumerate(attn_prefixes):
        attn[i] = _pack_attn(h, sd, p, C, H)
    for i in range(4):
        lat[i] = _pack_attn(h, sd, f"latent_conditioner.{i + 1}", C, H)
    w = E.DiffWeights()
    w.w_latent_conv = _p(h.conv(sd["latent_conditioner.0.weight"]))
    w.b_latent_conv = _p(h.f32(sd["latent_conditioner.0.bias"]))
    w.latent_attn_host = lat
    w.code_norm_g = _p(h.f32(sd["code_norm.weight"]))
    w.code_norm_b = _p(h.f32(sd["code_norm.bias"]))
    w.uncond_emb = _p(h.f32(sd["unconditioned_embedding"].reshape(-1)))
    w.w_time1 = _p(h.op(sd["time_embed.0.weight"]))
    w.b_time1 = _p(h.f32(sd["time_embed.0.bias"]))
    w.w_time2 = _p(h.op(sd["time_embed.2.weight"]))
    w.b_time2 = _p(h.f32(sd["time_embed.2.bias"]))
    w.w_emb_all = _p(h.op(torch.cat([sd[f"{p}.emb_layers.1.weight"] for p in res_prefixes], dim=0)))
    w.b_emb_all = _p(h.f32(torch.cat([sd[f"{p}.emb_layers.1.bias"] for p in res_prefixes], dim=0)))
    w.res_host = res
    w.attn_host = attn
    w.w_inp = _p(h.conv(sd["inp_block.weight"], in_pad))
    w.b_inp = _p(h.f32(sd["inp_block.bias"]))
    w.w_integ = _p(h.op(sd["integrating_conv.weight"]))
    w.b_integ = _p(h.f32(sd["integrating_conv.bias"]))
    w.out_gn_g = _p(h.f32(sd["out.0.weight"]))
    w.out_gn_b = _p(h.f32(sd["out.0.bias"]))
    w.w_final = _p(h.conv(sd["out.2.weight"]))
    w.b_final = _p(h.f32(sd["out.2.bias"]))
    h.keep += [res, attn, lat]
    h.weights = w
    h.in_pad = in_pad
    return h


# ----------------------------------------------------------------------------------------- conditioning front-end
def _pack_attn_natural(h, sd, prefix, C, heads):
    """AttentionBlock whose heads are not 64 wide: QKV rows stay in the reference's [head][q|k|v][ch] order (cond.hip's
    wave-per-query attention reads that layout); the relative-position table is scaled by sqrt(ch) (arch_util.py:106)."""
    a = E.AttnBlock()
    a.norm_g = _p(h.f32(sd[f"{prefix}.norm.weight"]))
    a.norm_b = _p(h.f32(sd[f"{prefix}.norm.bias"]))
    a.w_qkv = _p(h.op(sd[f"{prefix}.qkv.weight"]))
    a.b_qkv = _p(h.f32(sd[f"{prefix}.qkv.bias"]))
    a.w_proj = _p(h.op(sd[f"{prefix}.proj_out.weight"]))
    a.b_proj = _p(h.f32(sd[f"{prefix}.proj_out.bias"]))
    key = f"{prefix}.relative_pos_embeddings.relative_attention_bias.weight"
    a.relpos = _p(h.f32(relpos_table(sd[key], (C // heads) ** 0.5))) if key in sd else None
    return a


def pack_conditioning(sd_ar, sd_diff, ar_cfg: ARConfig, diff_cfg: DiffusionConfig, device, dtype, mel_pad=128):
    """Weights of UnifiedVoice.conditioning_encoder (autoregressive.py:204-228) and DiffusionTts.contextual_embedder
    (diffusion_decoder.py:186-192) for tt_cond_create."""
    h = Holder(device, dtype)
    D, Hh = ar_cfg.model_dim, ar_cfg.heads
    n_ar = 0
    while f"conditioning_encoder.attn.{n_ar}.norm.weight" in sd_ar:
        n_ar += 1
    ar_attn = (E.AttnBlock * n_ar)()
    for i in range(n_ar):
        p = f"conditioning_encoder.attn.{i}"
        ar_attn[i] = _pack_attn(h, sd_ar, p, D, Hh) if D // Hh == 64 else _pack_attn_natural(h, sd_ar, p, D, Hh)
    have_diff = sd_diff is not None  # the streaming path (api_fast) has no diffusion stage: autoregressive encoder only
    C_, Hd = (diff_cfg.model_channels, diff_cfg.num_heads) if have_diff else (64, 1)
    n_d = 0
    while have_diff and f"contextual_embedder.{2 + n_d}.norm.weight" in sd_diff:
        n_d += 1
    d_attn = (E.AttnBlock * max(n_d, 1))()
    for i in range(n_d):
        p = f"contextual_embedder.{2 + i}"
        d_attn[i] = _pack_attn(h, sd_diff, p, 2 * C_, Hd) if (2 * C_) // Hd == 64 else _pack_attn_natural(h, sd_diff, p, 2 * C_, Hd)
    w = E.CondWeights()
    wi = sd_ar["conditioning_encoder.init.weight"]
    n_mel_ar = wi.shape[1]
    wpad = torch.zeros(D, mel_pad)
    wpad[:, :n_mel_ar] = wi.detach().float().cpu().reshape(D, n_mel_ar)
    w.ar_w_init = _p(h.op(wpad))
    w.ar_b_init = _p(h.f32(sd_ar["conditioning_encoder.init.bias"]))
    w.ar_attn_host = ar_attn
    n_mel_d = 100
    if have_diff:
        n_mel_d = sd_diff["contextual_embedder.0.weight"].shape[1]
        w.diff_w_c0 = _p(h.conv(sd_diff["contextual_embedder.0.weight"], mel_pad))
        w.diff_b_c0 = _p(h.f32(sd_diff["contextual_embedder.0.bias"]))
        w.diff_w_c1 = _p(h.conv(sd_diff["contextual_embedder.1.weight"]))
        w.diff_b_c1 = _p(h.f32(sd_diff["contextual_embedder.1.bias"]))
    w.diff_attn_host = d_attn
    h.keep += [ar_attn, d_attn]
    h.weights = w
    h.shape = dict(ar_blocks=n_ar, diff_blocks=n_d, ar_mel=n_mel_ar, diff_mel=n_mel_d, mel_pad=mel_pad, diff_channels=C_, diff_heads=Hd)
    return h


# ----------------------------------------------------------------------------------------- HiFi-GAN decoder
def _cpad(c):
    return max(64, (c + 63) // 64 * 64)


def pack_hifigan(sd, cfg, device, dtype):
    """HifiganGenerator weights (weight norm already folded) for tt_hifi_create: token-major conv-GEMM operands with the
    channel widths padded to multiples of 64 by zero rows / columns; ConvTranspose1d(k = 2u, stride u) as the 2-tap GEMM
    with N = u * C_out described in csrc/hifigan.hip."""
    h = Holder(device, dtype)
    c0 = cfg.upsample_initial_channel
    ns, nk, nd = len(cfg.upsample_factors), len(cfg.resblock_kernel_sizes), len(cfg.resblock_dilation_sizes)
    assert ns <= E.HIFI_MAX_STAGES and nk <= 3 and nd <= 3

    def conv_padded(wt, cout_pad, cin_pad):
        wt = wt.detach().float().cpu()
        co, ci, k = wt.shape
        p = torch.zeros(cout_pad, k, cin_pad)
        p[:co, :, :ci] = wt.permute(0, 2, 1)
        return h.op(p)

    def bias_padded(b, n):
        p = torch.zeros(n)
        p[:b.numel()] = b.detach().float().cpu()
        return h.f32(p)

    w = E.HifiWeights()
    w.w_pre = _p(conv_padded(sd["conv_pre.weight"], _cpad(c0), cfg.in_channels))
    w.b_pre = _p(bias_padded(sd["conv_pre.bias"], _cpad(c0)))
    w.w_cond = _p(h.op(sd["cond_layer.weight"]))
    w.b_cond = _p(h.f32(sd["cond_layer.bias"]))
    res = (E.HifiResBlock * (ns * nk))()
    for i, (u, k) in enumerate(zip(cfg.upsample_factors, cfg.upsample_kernel_sizes)):
        assert k == 2 * u and u % 2 == 0, "transposed convs must have kernel = 2 * stride (api_fast.py:224)"
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        cin_p, cout_p = _cpad(cin), _cpad(cout)
        wt = sd[f"ups.{i}.weight"].detach().float().cpu()       # [cin][cout][2u]
        p = torch.zeros(u, cout_p, 2, cin_p)
        p[:, :cout, 0, :cin] = wt[:, :, u:].permute(2, 1, 0)      # tap 0 reads x[j - 1]: kernel index r + u
        p[:, :cout, 1, :cin] = wt[:, :, :u].permute(2, 1, 0)      # tap 1 reads x[j]:     kernel index r
        w.w_up[i] = _p(h.op(p.reshape(u * cout_p, 2 * cin_p)))
        b = torch.zeros(u, cout_p)
        b[:, :cout] = sd[f"ups.{i}.bias"].detach().float().cpu()[None]
        w.b_up[i] = _p(h.f32(b.reshape(-1)))
        for j in range(nk):
            rb = res[i * nk + j]
            for dd in range(nd):
                pre = f"resblocks.{i * nk + j}"
                rb.w1[dd] = _p(conv_padded(sd[f"{pre}.convs1.{dd}.weight"], cout_p, cout_p))
                rb.b1[dd] = _p(bias_padded(sd[f"{pre}.convs1.{dd}.bias"], cout_p))
                rb.w2[dd] = _p(conv_padded(sd[f"{pre}.convs2.{dd}.weight"], cout_p, cout_p))
                rb.b2[dd] = _p(bias_padded(sd[f"{pre}.convs2.{dd}.bias"], cout_p))
    c_last = _cpad(c0 // (2 ** ns))
    w.res_host = res
    w.w_post = _p(conv_padded(sd["conv_post.weight"], 1, c_last))
    w.b_post = _p(h.f32(sd["conv_post.bias"]))
    h.keep.append(res)
    h.weights = w
    return h


# ----------------------------------------------------------------------------------------- vocoder
def pack_vocoder(sd_folded, cfg: VocoderConfig, device, dtype, mel_pad=128):
    """sd_folded: UnivNet state_dict with weight-norm already folded (weights.fold_weight_norm)."""
    h = Holder(device, dtype)
    sd = sd_folded
    blocks = (E.VocBlock * len(cfg.strides))()
    for bi, stride in enumerate(cfg.strides):
        p = f"res_stack.{bi}"
        kp = f"{p}.kernel_predictor"
        b = blocks[bi]
        b.stride = stride
        b.w_convt = _p(h.f32(sd[f"{p}.convt_pre.1.weight"]))
        b.b_convt = _p(h.f32(sd[f"{p}.convt_pre.1.bias"]))
        b.w_kp_in = _p(h.conv(sd[f"{kp}.input_conv.0.weight"], mel_pad))
        b.b_kp_in = _p(h.f32(sd[f"{kp}.input_conv.0.bias"]))
        for r in range(3):
            for c, idx in enumerate((1, 3)):
                b.w_kp_res[2 * r + c] = _p(h.conv(sd[f"{kp}.residual_convs.{r}.{idx}.weight"]))
                b.b_kp_res[2 * r + c] = _p(h.f32(sd[f"{kp}.residual_convs.{r}.{idx}.bias"]))
        b.w_kp_kernel = _p(h.conv(sd[f"{kp}.kernel_conv.weight"]))
        b.b_kp_kernel = _p(h.f32(sd[f"{kp}.kernel_conv.bias"]))
        b.w_kp_bias = _p(h.conv(sd[f"{kp}.bias_conv.weight"]))
        b.b_kp_bias = _p(h.f32(sd[f"{kp}.bias_conv.bias"]))
        for j in range(len(cfg.dilations)):
            b.w_conv[j] = _p(h.f32(sd[f"{p}.conv_blocks.{j}.1.weight"]))
            b.b_conv[j] = _p(h.f32(sd[f"{p}.conv_blocks.{j}.1.bias"]))
    w = E.VocWeights()
    w.w_pre = _p(h.f32(sd["conv_pre.weight"]))
    w.b_pre = _p(h.f32(sd["conv_pre.bias"]))
    w.blocks_host = blocks
    w.w_post = _p(h.f32(sd["conv_post.1.weight"]))
    w.b_post = _p(h.f32(sd["conv_post.1.bias"]))
    h.keep.append(blocks)
    h.weights = w
    h.mel_pad = mel_pad
    return h
