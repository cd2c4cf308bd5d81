"""Reference state_dict layouts (key -> shape) and a seeded synthetic-weight generator.

The engine consumes checkpoints in the *reference's own* key layout
(autoregressive.pth / diffusion_decoder.pth / clvp2.pth / vocoder.pth['model_g'],
reference: tortoise/api.py:221-237), so a user can point it at the files the reference
downloads.  No checkpoints exist offline, so tests and bench.py build synthetic weights
with exactly these keys and shapes: every tensor is drawn from a seeded CPU
generator, including the tensors the reference zero-initialises (AttentionBlock.proj_out,
arch_util.py:111) — otherwise every attention branch would contribute exactly 0 and hide bugs.

`tests/test_manifest.py` checks these manifests against the live reference modules when
/root/reference is present.
"""
from collections import OrderedDict
import math
import torch

from .config import ARConfig, DiffusionConfig, CLVPConfig, CVVPConfig, VocoderConfig


# ----------------------------------------------------------------------------- manifests
def _attention_block(prefix, ch, heads, rel_pos=True):
    # arch_util.py:80-123
    d = OrderedDict()
    d[f"{prefix}.norm.weight"] = (ch,)
    d[f"{prefix}.norm.bias"] = (ch,)
    d[f"{prefix}.qkv.weight"] = (3 * ch, ch, 1)
    d[f"{prefix}.qkv.bias"] = (3 * ch,)
    d[f"{prefix}.proj_out.weight"] = (ch, ch, 1)
    d[f"{prefix}.proj_out.bias"] = (ch,)
    if rel_pos:
        d[f"{prefix}.relative_pos_embeddings.relative_attention_bias.weight"] = (32, heads)
    return d


def _res_block(prefix, ch):
    # diffusion_decoder.py:60-120 with use_scale_shift_norm=True, efficient_config=True, kernel 3
    d = OrderedDict()
    d[f"{prefix}.in_layers.0.weight"] = (ch,)
    d[f"{prefix}.in_layers.0.bias"] = (ch,)
    d[f"{prefix}.in_layers.2.weight"] = (ch, ch, 1)
    d[f"{prefix}.in_layers.2.bias"] = (ch,)
    d[f"{prefix}.emb_layers.1.weight"] = (2 * ch, ch)
    d[f"{prefix}.emb_layers.1.bias"] = (2 * ch,)
    d[f"{prefix}.out_layers.0.weight"] = (ch,)
    d[f"{prefix}.out_layers.0.bias"] = (ch,)
    d[f"{prefix}.out_layers.3.weight"] = (ch, ch, 3)
    d[f"{prefix}.out_layers.3.bias"] = (ch,)
    return d


def ar_manifest(cfg: ARConfig = ARConfig()):
    """UnifiedVoice.state_dict() (autoregressive.py:293-357).  Conditioning encoder keys are
    listed (they are in the checkpoint) but the engine never reads them (once-per-voice, out of scope)."""
    D = cfg.model_dim
    d = OrderedDict()
    d["conditioning_encoder.init.weight"] = (D, 80, 1)
    d["conditioning_encoder.init.bias"] = (D,)
    for i in range(6):
        d.update(_attention_block(f"conditioning_encoder.attn.{i}", D, cfg.heads, rel_pos=False))
    d["text_embedding.weight"] = (cfg.text_vocab, D)
    d["mel_embedding.weight"] = (cfg.number_mel_codes, D)
    for i in range(cfg.layers):
        p = f"gpt.h.{i}"
        d[f"{p}.ln_1.weight"] = (D,)
        d[f"{p}.ln_1.bias"] = (D,)
        d[f"{p}.attn.c_attn.weight"] = (D, 3 * D)  # HF Conv1D: [in, out]
        d[f"{p}.attn.c_attn.bias"] = (3 * D,)
        d[f"{p}.attn.c_proj.weight"] = (D, D)
        d[f"{p}.attn.c_proj.bias"] = (D,)
        d[f"{p}.ln_2.weight"] = (D,)
        d[f"{p}.ln_2.bias"] = (D,)
        d[f"{p}.mlp.c_fc.weight"] = (D, 4 * D)
        d[f"{p}.mlp.c_fc.bias"] = (4 * D,)
        d[f"{p}.mlp.c_proj.weight"] = (4 * D, D)
        d[f"{p}.mlp.c_proj.bias"] = (D,)
    d["gpt.ln_f.weight"] = (D,)
    d["gpt.ln_f.bias"] = (D,)
    d["mel_pos_embedding.emb.weight"] = (cfg.mel_pos_len, D)
    d["text_pos_embedding.emb.weight"] = (cfg.text_pos_len, D)
    d["final_norm.weight"] = (D,)
    d["final_norm.bias"] = (D,)
    d["text_head.weight"] = (cfg.text_vocab, D)
    d["text_head.bias"] = (cfg.text_vocab,)
    d["mel_head.weight"] = (cfg.number_mel_codes, D)
    d["mel_head.bias"] = (cfg.number_mel_codes,)
    return d


def diffusion_manifest(cfg: DiffusionConfig = DiffusionConfig()):
    """DiffusionTts.state_dict() (diffusion_decoder.py:134-210)."""
    C, H = cfg.model_channels, cfg.num_heads
    d = OrderedDict()
    d["unconditioned_embedding"] = (1, C, 1)
    d["inp_block.weight"] = (C, cfg.in_channels, 3)
    d["inp_block.bias"] = (C,)
    for i in (0, 2):
        d[f"time_embed.{i}.weight"] = (C, C)
        d[f"time_embed.{i}.bias"] = (C,)
    d["code_embedding.weight"] = (cfg.in_tokens, C)
    for i in range(3):
        d.update(_attention_block(f"code_converter.{i}", C, H))
    d["code_norm.weight"] = (C,)
    d["code_norm.bias"] = (C,)
    d["latent_conditioner.0.weight"] = (C, cfg.in_latent_channels, 3)
    d["latent_conditioner.0.bias"] = (C,)
    for i in range(1, 5):
        d.update(_attention_block(f"latent_conditioner.{i}", C, H))
    d["contextual_embedder.0.weight"] = (C, cfg.in_channels, 3)
    d["contextual_embedder.0.bias"] = (C,)
    d["contextual_embedder.1.weight"] = (2 * C, C, 3)
    d["contextual_embedder.1.bias"] = (2 * C,)
    for i in range(2, 7):
        d.update(_attention_block(f"contextual_embedder.{i}", 2 * C, H))
    for i in range(3):
        d.update(_res_block(f"conditioning_timestep_integrator.{i}.resblk", C))
        d.update(_attention_block(f"conditioning_timestep_integrator.{i}.attn", C, H))
    d["integrating_conv.weight"] = (C, 2 * C, 1)
    d["integrating_conv.bias"] = (C,)
    d["mel_head.weight"] = (cfg.in_channels, C, 3)
    d["mel_head.bias"] = (cfg.in_channels,)
    for i in range(cfg.num_layers):
        d.update(_res_block(f"layers.{i}.resblk", C))
        d.update(_attention_block(f"layers.{i}.attn", C, H))
    for i in range(cfg.num_layers, cfg.num_layers + 3):
        d.update(_res_block(f"layers.{i}", C))
    d["out.0.weight"] = (C,)
    d["out.0.bias"] = (C,)
    d["out.2.weight"] = (cfg.out_channels, C, 3)
    d["out.2.bias"] = (cfg.out_channels,)
    return d


def clvp_manifest(cfg: CLVPConfig = CLVPConfig()):
    """CLVP.state_dict() with use_xformers=True (clvp.py:47-97; xtransformers.py:731-904)."""
    D = cfg.dim
    inner = D * cfg.ff_mult
    d = OrderedDict()
    d["temperature"] = ()
    d["text_emb.weight"] = (cfg.num_text_tokens, D)
    d["to_text_latent.weight"] = (cfg.dim_latent, D)
    d["speech_emb.weight"] = (cfg.num_speech_tokens, D)
    d["to_speech_latent.weight"] = (cfg.dim_latent, D)
    for tower in ("text_transformer", "speech_transformer"):
        base = f"{tower}.transformer"
        for li in range(2 * cfg.depth):
            p = f"{base}.attn_layers.layers.{li}"
            d[f"{p}.0.0.g"] = (D,)
            if li % 2 == 0:  # attention
                for nm in ("to_q", "to_k", "to_v"):
                    d[f"{p}.1.wrap.{nm}.weight"] = (D, D)
                d[f"{p}.1.wrap.to_out.weight"] = (D, D)
                d[f"{p}.1.wrap.to_out.bias"] = (D,)
            else:  # GEGLU feed-forward
                d[f"{p}.1.wrap.net.0.proj.weight"] = (2 * inner, D)
                d[f"{p}.1.wrap.net.0.proj.bias"] = (2 * inner,)
                d[f"{p}.1.wrap.net.3.weight"] = (D, inner)
                d[f"{p}.1.wrap.net.3.bias"] = (D,)
        d[f"{base}.attn_layers.rotary_pos_emb.inv_freq"] = (cfg.rotary_dim // 2,)
        d[f"{base}.norm.weight"] = (D,)
        d[f"{base}.norm.bias"] = (D,)
    return d


def cvvp_manifest(cfg: CVVPConfig = CVVPConfig()):
    """CVVP.state_dict() with mel_codes set (cvvp.py:63-98; CollapsingTransformer 19-51; xtransformers.py:731-904, 1187-1213)."""
    D, L = cfg.model_dim, cfg.latent_dim
    d = OrderedDict()
    d["temperature"] = ()
    d["cond_emb.0.weight"] = (D // 2, cfg.mel_channels, 5)
    d["cond_emb.0.bias"] = (D // 2,)
    d["cond_emb.1.weight"] = (D, D // 2, 3)
    d["cond_emb.1.bias"] = (D,)
    for tower, out in (("conditioning_transformer", D), ("speech_transformer", L)):
        base = f"{tower}.transformer"
        d[f"{base}.attn_layers.rotary_pos_emb.inv_freq"] = (cfg.rotary_dim // 2,)
        for li in range(2 * cfg.depth):
            p = f"{base}.attn_layers.layers.{li}"
            d[f"{p}.0.0.g"] = (D,)
            if li % 2 == 0:
                for nm in ("to_q", "to_k", "to_v"):
                    d[f"{p}.1.{nm}.weight"] = (D, D)
                d[f"{p}.1.to_out.weight"] = (D, D)
                d[f"{p}.1.to_out.bias"] = (D,)
            else:  # GEGLU feed-forward, ff_mult = 1
                d[f"{p}.1.net.0.proj.weight"] = (2 * D, D)
                d[f"{p}.1.net.0.proj.bias"] = (2 * D,)
                d[f"{p}.1.net.3.weight"] = (D, D)
                d[f"{p}.1.net.3.bias"] = (D,)
        d[f"{base}.norm.weight"] = (D,)
        d[f"{base}.norm.bias"] = (D,)
        d[f"{tower}.pre_combiner.0.weight"] = (out, D, 1)
        d[f"{tower}.pre_combiner.0.bias"] = (out,)
        d.update(_attention_block(f"{tower}.pre_combiner.1", out, cfg.heads, rel_pos=False))
        d[f"{tower}.pre_combiner.2.weight"] = (out, out, 1)
        d[f"{tower}.pre_combiner.2.bias"] = (out,)
        if tower == "conditioning_transformer":
            d["to_conditioning_latent.weight"] = (L, L)
            d["speech_emb.emb.weight"] = (cfg.mel_codes, D)
    d["to_speech_latent.weight"] = (L, L)
    return d


def _wn(d, prefix, shape, g_dim0=None):
    d[f"{prefix}.bias"] = (shape[0],) if g_dim0 is None else (g_dim0,)
    d[f"{prefix}.weight_g"] = (shape[0], 1, 1)
    d[f"{prefix}.weight_v"] = tuple(shape)


def vocoder_manifest(cfg: VocoderConfig = VocoderConfig()):
    """UnivNetGenerator.state_dict() before remove_weight_norm (vocoder.py:225-265, 7-64, 104-153).
    weight_norm stores weight_g [dim0,1,1] and weight_v; the engine folds them at load."""
    c = cfg.channel_size
    hid = cfg.kpnet_hidden
    nl = len(cfg.dilations)
    d = OrderedDict()
    for bi, stride in enumerate(cfg.strides):
        p = f"res_stack.{bi}"
        kp = f"{p}.kernel_predictor"
        _wn(d, f"{kp}.input_conv.0", (hid, cfg.n_mel_channels, 5))
        for r in range(3):
            _wn(d, f"{kp}.residual_convs.{r}.1", (hid, hid, cfg.kpnet_conv_size))
            _wn(d, f"{kp}.residual_convs.{r}.3", (hid, hid, cfg.kpnet_conv_size))
        _wn(d, f"{kp}.kernel_conv", (c * 2 * c * 3 * nl, hid, cfg.kpnet_conv_size))
        _wn(d, f"{kp}.bias_conv", (2 * c * nl, hid, cfg.kpnet_conv_size))
        _wn(d, f"{p}.convt_pre.1", (c, c, 2 * stride))  # ConvTranspose1d: [in, out, k], bias [out]
        for j in range(nl):
            _wn(d, f"{p}.conv_blocks.{j}.1", (c, c, 3))
    _wn(d, "conv_pre", (c, cfg.noise_dim, 7))
    _wn(d, "conv_post.1", (1, c, 7))
    return d


def hifigan_manifest(cfg):
    """HifiganGenerator.state_dict() with weight_norm parameters (hifigan_decoder.py:191-228): conv_pre, ups.i
    (ConvTranspose1d: [in, out, k]), resblocks.(i * n_kernels + j).convs1/convs2.d, conv_post, cond_layer (plain Conv1d)."""
    d = OrderedDict()
    c0 = cfg.upsample_initial_channel
    _wn(d, "conv_pre", (c0, cfg.in_channels, 7))
    nk = len(cfg.resblock_kernel_sizes)
    ch = c0
    for i, (u, k) in enumerate(zip(cfg.upsample_factors, cfg.upsample_kernel_sizes)):
        cin, ch = c0 // (2 ** i), c0 // (2 ** (i + 1))
        _wn(d, f"ups.{i}", (cin, ch, k), g_dim0=ch)
        for j, ks in enumerate(cfg.resblock_kernel_sizes):
            for dd in range(len(cfg.resblock_dilation_sizes)):
                _wn(d, f"resblocks.{i * nk + j}.convs1.{dd}", (ch, ch, ks))
                _wn(d, f"resblocks.{i * nk + j}.convs2.{dd}", (ch, ch, ks))
    _wn(d, "conv_post", (1, ch, 7))
    d["cond_layer.weight"] = (c0, cfg.cond_channels, 1)
    d["cond_layer.bias"] = (c0,)
    return d


def rlg_manifest(channels):
    """RandomLatentConverter(channels) (random_latent_generator.py:42-55): 5 EqualLinear + 1 Linear; rlg_auto.pth is
    channels = 1024, rlg_diffuser.pth 2048 (api.py:301-309)."""
    m = OrderedDict()
    for i in range(6):
        m[f"layers.{i}.weight"] = (channels, channels)
        m[f"layers.{i}.bias"] = (channels,)
    return m


# ----------------------------------------------------------------------------- synthetic weights
def _is_norm_gain(key):
    k = key
    return (k.endswith(".g") or ".ln_" in k or ".ln_f." in k or "final_norm" in k or ".norm." in k
            or "code_norm" in k or ".in_layers.0." in k or ".out_layers.0." in k or k.startswith("out.0.")
            or k.endswith("transformer.norm.weight"))


def synthetic_state_dict(manifest, seed, gain=1.0):
    """Seeded, reference-independent weights for a manifest.  Deterministic for a given torch
    build (CPU mt19937 generator), which both the build container and the GPU box share."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    sd = OrderedDict()
    for key, shape in manifest.items():
        if key.endswith("inv_freq"):
            n = shape[0]
            sd[key] = 1.0 / (10000 ** (torch.arange(0, 2 * n, 2).float() / (2 * n)))
            continue
        if key == "temperature":
            sd[key] = torch.tensor(1.0)
            continue
        t = torch.randn(shape, generator=g, dtype=torch.float32)
        if key.endswith("relative_attention_bias.weight"):
            t = t * 0.5
        elif key.endswith("weight_g"):
            # folded weight rows get norm == g: keep UnivNet activations O(1) under random weights
            # (mel inputs have RMS ~6; an LVC output sums 96 products)
            base = 1.0
            if "input_conv" in key:
                base = 0.15
            elif "kernel_conv" in key:
                base = 0.1
            elif "conv_post" in key:
                base = 0.3
            t = base * (1.0 + 0.1 * t)
        elif key.endswith("weight_v"):
            t = t * 0.05
        elif key == "unconditioned_embedding":
            pass
        elif "embedding" in key or key.endswith("_emb.weight") or ".emb." in key:
            t = t * 0.05
        elif len(shape) == 1:
            if key.endswith("bias"):
                t = t * 0.05
            elif _is_norm_gain(key):
                t = 1.0 + 0.1 * t
            else:
                t = t * 0.05
        else:
            # dense / conv weights: fan-in scaling.  HF Conv1D stores [in, out].
            if ".attn.c_" in key or ".mlp.c_" in key:
                fan_in = shape[0]
            else:
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
            t = t * (gain / math.sqrt(fan_in))
        sd[key] = t.contiguous()
    return sd


def fold_weight_norm(sd):
    """weight = g * v / ||v||, norm over all dims but 0 (torch.nn.utils.weight_norm, dim=0),
    which is what UnivNetGenerator.eval(inference=True) bakes in (vocoder.py:284-298)."""
    out = OrderedDict()
    for k, v in sd.items():
        if k.endswith("weight_v"):
            base = k[: -len("weight_v")]
            g = sd[base + "weight_g"]
            nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            out[base + "weight"] = (g * v / nrm).contiguous()
        elif k.endswith("weight_g"):
            continue
        else:
            out[k] = v
    return out


def suppress_stop_token(ar_sd, cfg: ARConfig = ARConfig(), value=-1e9):
    """Synthetic weights never emit a meaningful stop token, so fixed-length benchmarks and parity
    runs pin the stop logit far below everything else.  Applied identically to the
    oracle and the engine because both read the same state_dict."""
    sd = OrderedDict(ar_sd)
    b = sd["mel_head.bias"].clone()
    b[cfg.stop_mel_token] = value
    sd["mel_head.bias"] = b
    return sd
