"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain fp32 PyTorch / numpy) of the reference's
hot path.  It is the *checker* for the HIP engine: only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import it.  The product (tortoise_tts_amd) never does.

Every function states the reference lines it follows (paths relative to /root/reference).
Pinned by: tests/test_oracle_vs_reference.py (live comparison against the reference's own
nn.Modules, run in the build container) and tests/test_oracle_golden.py (committed golden
vectors produced from the reference's modules by oracle/make_golden.py).  The reference holds
no numeric golden vectors of its own, so that is the strongest pin available.

The AR sampling loop is a restatement of HF transformers `GenerationMixin.sample` (third-party; the
reference pins ==4.31.0 in setup.py:30, not vendored) following the in-repo fork
tortoise/models/stream_generator.py:916-1000 and the logits processors.  It is pinned against a REAL
`generate(do_sample=True, ...)` run of the reference's GPT2InferenceModel through the reference's own
`UnifiedVoice.inference_speech`, with the INSTALLED transformers (5.15) GenerationMixin mixed back in
(oracle/ref_shims.enable_generate): identical codes bit for bit for both position rules, ragged stop
rows and whole-batch early exit (tests/test_oracle_vs_reference.py::test_sampling_loop_equals_hf_generate,
committed as tests/golden/sampling.npz), AND against the reference's own in-tree copy of the 4.31 loop,
NewGenerationMixin.sample_stream (stream_generator.py:722-1000, the generator api_fast.py iterates), run live on the reference
model: identical codes in the same six cases, and its per-step latents equal ar_latents() of those codes
(tests/test_oracle_vs_reference.py::test_sampling_loop_and_streamed_latents_equal_reference_sample_stream).  What stays
unpinned: the processor / warper CLASSES come from the installed 5.15 in both pins (4.31.0 cannot be installed offline); their
semantics for these options equal 4.31's by inspection of the processors' documented behaviour.

All state is passed as reference-layout state_dicts (see tortoise_tts_amd/weights.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from tortoise_tts_amd.config import (ARConfig, DiffusionConfig, CLVPConfig, CVVPConfig, VocoderConfig, CALM_TOKEN,
                                     TACOTRON_MEL_MAX, TACOTRON_MEL_MIN)


# =============================================================================== GPT-2 trunk
def gelu_new(x):
    # HF "gelu_new" (tanh approximation), GPT2Config default activation_function.
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def gpt2_block(sd, i, x, heads, past=None, causal_offset=0):
    """One HF GPT2Block (pre-LN).  x: [B, n, D].  past: (k, v) each [B, H, ctx, hd] or None.
    Returns (x_out, (k_all, v_all)).  HF Conv1D computes x @ W + b with W stored [in, out]."""
    p = f"gpt.h.{i}"
    B, n, D = x.shape
    hd = D // heads
    h = F.layer_norm(x, (D,), sd[f"{p}.ln_1.weight"], sd[f"{p}.ln_1.bias"], 1e-5)
    qkv = h @ sd[f"{p}.attn.c_attn.weight"] + sd[f"{p}.attn.c_attn.bias"]
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(B, n, heads, hd).transpose(1, 2)
    k = k.view(B, n, heads, hd).transpose(1, 2)
    v = v.view(B, n, heads, hd).transpose(1, 2)
    if past is not None:
        k = torch.cat([past[0], k], dim=2)
        v = torch.cat([past[1], v], dim=2)
    ctx = k.shape[2]
    att = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    # causal: query j (absolute position ctx-n+j) sees keys <= its position
    qpos = torch.arange(ctx - n, ctx)[:, None]
    kpos = torch.arange(ctx)[None, :]
    att = att.masked_fill(kpos > qpos, torch.finfo(att.dtype).min)
    att = torch.softmax(att, dim=-1)
    a = (att @ v).transpose(1, 2).reshape(B, n, D)
    a = a @ sd[f"{p}.attn.c_proj.weight"] + sd[f"{p}.attn.c_proj.bias"]
    x = x + a
    h = F.layer_norm(x, (D,), sd[f"{p}.ln_2.weight"], sd[f"{p}.ln_2.bias"], 1e-5)
    h = gelu_new(h @ sd[f"{p}.mlp.c_fc.weight"] + sd[f"{p}.mlp.c_fc.bias"])
    h = h @ sd[f"{p}.mlp.c_proj.weight"] + sd[f"{p}.mlp.c_proj.bias"]
    return x + h, (k, v)


def gpt2_trunk(sd, cfg: ARConfig, emb, past=None):
    """HF GPT2Model.forward with wte deleted and wpe == zeros (autoregressive.py:260-264):
    30 blocks then ln_f.  Returns (hidden [B,n,D], presents)."""
    x = emb
    presents = []
    for i in range(cfg.layers):
        x, kv = gpt2_block(sd, i, x, cfg.heads, None if past is None else past[i])
        presents.append(kv)
    D = cfg.model_dim
    x = F.layer_norm(x, (D,), sd["gpt.ln_f.weight"], sd["gpt.ln_f.bias"], 1e-5)
    return x, presents


def ar_head(sd, cfg: ARConfig, hidden):
    """lm_head = Sequential(final_norm, mel_head) (autoregressive.py:42, 174)."""
    D = cfg.model_dim
    h = F.layer_norm(hidden, (D,), sd["final_norm.weight"], sd["final_norm.bias"], 1e-5)
    return h @ sd["mel_head.weight"].t() + sd["mel_head.bias"]


def ar_prefix(sd, cfg: ARConfig, cond_latent, text_tokens):
    """UnifiedVoice.inference_speech prefix (autoregressive.py:538-544).
    cond_latent [1, D]; text_tokens int [1, T] (already F.pad'ed by api.py:391).
    Returns emb [1, P, D] with P = 1 + T + 2."""
    t = F.pad(text_tokens.long(), (0, 1), value=cfg.stop_text_token)
    t = F.pad(t, (1, 0), value=cfg.start_text_token)
    n = t.shape[1]
    text_emb = sd["text_embedding.weight"][t] + sd["text_pos_embedding.emb.weight"][:n][None]
    return torch.cat([cond_latent[:, None, :], text_emb], dim=1)


def ar_prefill(sd, cfg: ARConfig, prefix_emb, batch):
    """First generate() step: GPT2InferenceModel.forward prefill branch (autoregressive.py:134-144):
    [prefix ‖ mel_embedding(start) + mel_pos[0]] for each of `batch` rows.
    Returns (logits_last [B, V], presents)."""
    start = sd["mel_embedding.weight"][cfg.start_mel_token] + sd["mel_pos_embedding.emb.weight"][0]
    emb = torch.cat([prefix_emb, start[None, None, :]], dim=1).repeat(batch, 1, 1)
    hidden, presents = gpt2_trunk(sd, cfg, emb)
    return ar_head(sd, cfg, hidden[:, -1]), presents


def ar_mel_position(step_index, kv_cache=True):
    """Mel position-embedding row used for the token fed at decode step `step_index` (>= 1; the
    start token, index 0, is handled by ar_prefill and uses row 0).
    kv_cache=True path (autoregressive.py:145-149): row = attention_mask.shape[1] - mel_len
    = (P + 1 + step_index) - P = step_index + 1   -> 0, 2, 3, 4, ...   
    kv_cache=False re-runs the prefill branch each step: row = step_index."""
    return step_index + 1 if kv_cache else step_index


def ar_step(sd, cfg: ARConfig, tokens, step_index, past, kv_cache=True):
    """One cached decode step (autoregressive.py:145-163).  tokens int64 [B]."""
    pos = ar_mel_position(step_index, kv_cache)
    emb = sd["mel_embedding.weight"][tokens] + sd["mel_pos_embedding.emb.weight"][pos][None]
    hidden, presents = gpt2_trunk(sd, cfg, emb[:, None, :], past)
    return ar_head(sd, cfg, hidden[:, -1]), presents


# =============================================================================== HF 4.31 sampling
def repetition_penalty_(scores, input_ids, penalty):
    """RepetitionPenaltyLogitsProcessor (transformers 4.31): gather / where(<0, *p, /p) / scatter,
    over ALL input ids including the fake prefix ids {1, start_mel_token}."""
    s = torch.gather(scores, 1, input_ids)
    s = torch.where(s < 0, s * penalty, s / penalty)
    return scores.scatter(1, input_ids, s)


def top_k_(scores, k):
    """TopKLogitsWarper (4.31): remove everything strictly below the k-th largest value."""
    k = min(k, scores.shape[-1])
    kth = torch.topk(scores, k)[0][..., -1, None]
    return scores.masked_fill(scores < kth, -float("inf"))


def top_p_(scores, top_p, min_tokens_to_keep=1):
    """TopPLogitsWarper (4.31): ascending sort, drop tokens whose cumulative prob <= 1 - top_p."""
    sorted_logits, sorted_idx = torch.sort(scores, descending=False)
    cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
    remove = cum <= (1 - top_p)
    remove[..., -min_tokens_to_keep:] = False
    remove = remove.scatter(1, sorted_idx, remove)
    return scores.masked_fill(remove, -float("inf"))


def typical_(scores, mass=0.9):
    """TypicalLogitsWarper (tortoise/utils/typical_sampling.py:11-33; `typical_sampling=True` of tts(), api.py:361-364,
    autoregressive.py:558): tokens ordered by |surprisal - entropy| ascending; the shortest such prefix whose probability reaches
    `mass` stays (the token that crosses the mass included, ties at its distance included), everything farther from the entropy is removed."""
    logp = torch.log_softmax(scores, dim=-1)
    p = logp.exp()
    entropy = -(logp * p).nansum(-1, keepdim=True)  # 0 * -inf of a suppressed token is skipped
    dist = ((-logp) - entropy).abs()
    dist_sorted, order = torch.sort(dist, descending=False)
    cum = scores.gather(-1, order).softmax(dim=-1).cumsum(dim=-1)
    last = (cum < mass).sum(dim=1).clamp_(min=0)
    remove_sorted = dist_sorted > dist_sorted.gather(1, last.view(-1, 1))
    remove = remove_sorted.scatter(1, order, remove_sorted)
    return scores.masked_fill(remove, -float("inf"))


def warp_logits(logits, input_ids, repetition_penalty=2.0, temperature=0.8, top_k=50, top_p=0.8, typical_mass=None):
    """Processor order in 4.31 `generate(do_sample=True)`: repetition penalty (processor), the caller's logits_processor list
    (inference_speech passes [TypicalLogitsWarper(typical_mass)] when typical_sampling=True, autoregressive.py:558: generate()
    appends it to the default processors), then warpers temperature -> top-k (GenerationConfig default 50, never overridden by
    api.py) -> top-p."""
    s = logits.float()
    if repetition_penalty is not None and repetition_penalty != 1.0:
        s = repetition_penalty_(s, input_ids, repetition_penalty)
    if typical_mass is not None:
        s = typical_(s, typical_mass)
    if temperature is not None and temperature != 1.0:
        s = s / temperature
    if top_k is not None and top_k != 0:
        s = top_k_(s, top_k)
    if top_p is not None and top_p < 1.0:
        s = top_p_(s, top_p)
    return s


def multinomial_from_exponential(probs, q):
    """torch.multinomial(probs, 1) on CPU == argmax(probs / q) with q ~ Exp(1) drawn from the same
    generator state (probed in SURVEY.md §8c).  q is an *input* so oracle and engine share noise."""
    return torch.argmax(probs / q, dim=-1)


def ar_sample_loop(sd, cfg: ARConfig, cond_latent, text_tokens, batch, max_new, exp_noise,
                   repetition_penalty=2.0, temperature=0.8, top_k=50, top_p=0.8, kv_cache=True,
                   return_logits=False, typical_mass=None):
    """UnifiedVoice.inference_speech + GenerationMixin.sample (autoregressive.py:535-563;
    stream_generator.py:916-1000).  exp_noise: [max_new, batch, V] Exp(1) draws.
    Returns int64 codes [batch, n] (n <= max_new; shorter only if every row hit stop), like
    `gen[:, trunc_index:]`."""
    prefix = ar_prefix(sd, cfg, cond_latent, text_tokens)
    P = prefix.shape[1]
    # fake_inputs (autoregressive.py:546-548): P ones + start_mel_token
    input_ids = torch.full((batch, P + 1), 1, dtype=torch.long)
    input_ids[:, -1] = cfg.start_mel_token
    unfinished = torch.ones(batch, dtype=torch.long)
    logits, past = ar_prefill(sd, cfg, prefix, batch)
    all_logits = []
    for step in range(max_new):
        if return_logits:
            all_logits.append(logits.clone())
        scores = warp_logits(logits, input_ids, repetition_penalty, temperature, top_k, top_p, typical_mass)
        probs = torch.softmax(scores, dim=-1)
        nxt = multinomial_from_exponential(probs, exp_noise[step])
        nxt = nxt * unfinished + cfg.stop_mel_token * (1 - unfinished)
        input_ids = torch.cat([input_ids, nxt[:, None]], dim=1)
        unfinished = unfinished * (nxt != cfg.stop_mel_token).long()
        if unfinished.max() == 0 or step == max_new - 1:
            break
        logits, past = ar_step(sd, cfg, nxt, step + 1, past, kv_cache)
    codes = input_ids[:, P + 1:]
    if return_logits:
        return codes, torch.stack(all_logits)
    return codes


# =============================================================================== random-voice latents
def random_latent_converter(sd, r, lr_mul=0.1):
    """RandomLatentConverter.forward with the Gaussian input `r` [B, C] injected (random_latent_generator.py:8-55):
    5 x EqualLinear (F.linear(x, W * scale), scale = lr_mul / sqrt(C); leaky_relu(. + b * lr_mul, 0.2) * sqrt(2)) and one
    plain Linear.  api.py:301-309 calls it with C = 1024 (rlg_auto) and 2048 (rlg_diffuser)."""
    x = r.float()
    C = x.shape[1]
    scale = (1.0 / math.sqrt(C)) * lr_mul
    for i in range(5):
        x = F.linear(x, sd[f"layers.{i}.weight"] * scale)
        x = F.leaky_relu(x + (sd[f"layers.{i}.bias"] * lr_mul)[None], negative_slope=0.2) * (2 ** 0.5)
    return F.linear(x, sd["layers.5.weight"], sd["layers.5.bias"])


# =============================================================================== integer post-processing
def fix_autoregressive_output(codes, stop_token=8193):
    """api.py:87-114 on one row (numpy int64 [n]); returns a new array.  Integer path: bit-exact."""
    codes = np.array(codes, dtype=np.int64, copy=True)
    idx = np.nonzero(codes == stop_token)[0]
    if len(idx) == 0:
        return codes
    codes[idx] = 83
    stm = int(idx.min())
    codes[stm:] = 83
    if stm - 3 < codes.shape[0]:
        codes[-3] = 45
        codes[-2] = 45
        codes[-1] = 248
    return codes


def pad_codes(codes, max_mel_tokens=500, stop_token=8193):
    """api.py:425-426."""
    codes = np.asarray(codes, dtype=np.int64)
    pad = max_mel_tokens - codes.shape[1]
    return np.pad(codes, ((0, 0), (0, pad)), constant_values=stop_token)


def calm_trim_length(codes_row, calm_token=CALM_TOKEN):
    """api.py:547-556: first k at which more than 8 consecutive calm tokens have been seen;
    latents are cut to [:k].  Returns len(codes_row) when no such run exists."""
    c = 0
    for k, v in enumerate(np.asarray(codes_row)):
        c = c + 1 if v == calm_token else 0
        if c > 8:
            return k
    return len(codes_row)


def topk_indices(scores, k):
    """api.py:477 torch.topk(clip_results, k).indices with the tie-break fixed to lowest index
     so every rank agrees."""
    s = np.asarray(scores, dtype=np.float64)
    order = np.lexsort((np.arange(len(s)), -s))
    return order[:k]


# =============================================================================== AR latent re-pass
def ar_latents(sd, cfg: ARConfig, cond_latent, text_tokens, codes, stream_positions=False):
    """UnifiedVoice.forward(return_latent=True, clip_inputs=False) (autoregressive.py:454-506, 417-431)
    as api.py:521-524 calls it: wav_lengths = n * mel_length_compression so set_mel_padding is a no-op.
    cond_latent [k, D]; text_tokens int [k, T]; codes int64 [k, n].  Returns [k, n, D].

    stream_positions: the latents the STREAMING path collects (api_fast.py:389-414: `final_norm(hidden_states[-1][:, -1])` of
    every sampling step, stream_generator.py:980) under kv_cache=True, where the cached decode feeds mel input j > 0 with position
    j + 1 (autoregressive.py:134-149: attention_mask.shape[1] - mel_len).  Causal attention makes those per-step states equal
    to one full pass whose mel positions are 0, 2, 3, ...; with kv_cache=False they equal the plain pass (positions 0, 1, 2, ...).
    Pinned live against the reference's own sample_stream (tests/test_oracle_vs_reference.py)."""
    k, n = codes.shape
    t = F.pad(text_tokens.long(), (0, 1), value=cfg.stop_text_token)
    t = F.pad(t, (1, 0), value=cfg.start_text_token)
    text_emb = sd["text_embedding.weight"][t] + sd["text_pos_embedding.emb.weight"][: t.shape[1]][None]
    m = F.pad(codes.long(), (0, 1), value=cfg.stop_mel_token)
    m = F.pad(m, (1, 0), value=cfg.start_mel_token)
    pos = torch.arange(m.shape[1])
    if stream_positions:
        pos = torch.where(pos > 0, pos + 1, pos)
    mel_emb = sd["mel_embedding.weight"][m] + sd["mel_pos_embedding.emb.weight"][pos][None]
    emb = torch.cat([cond_latent[:, None, :], text_emb, mel_emb], dim=1)
    hidden, _ = gpt2_trunk(sd, cfg, emb)
    enc = hidden[:, 1:]
    D = cfg.model_dim
    enc = F.layer_norm(enc, (D,), sd["final_norm.weight"], sd["final_norm.bias"], 1e-5)
    return enc[:, -m.shape[1]:][:, :-2]


# =============================================================================== CLVP
def _rmsnorm(x, g, eps=1e-8):
    # xtransformers.py:335-344
    norm = torch.norm(x, dim=-1, keepdim=True) * (x.shape[-1] ** -0.5)
    return x / norm.clamp(min=eps) * g


def _rotate_half(x):
    # xtransformers.py:277-280
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


def clvp_encoder(sd, cfg: CLVPConfig, tower, x, wrap=".wrap"):
    """x-transformers Encoder as CLVP builds it (clvp.py:54-83; xtransformers.py:731-1013):
    pre-RMSNorm, bias-free q/k/v, rotary on the first 32 dims of q, k AND v (625-629), softmax(q k^T / 8),
    to_out; GEGLU feed-forward with erf-GELU (429-437), final LayerNorm (1234).  Eval masks are all-ones.
    wrap: CLVP's CheckpointedXTransformerEncoder wraps every sublayer (state_dict keys `...layers.N.1.wrap.to_q`); CVVP's plain
    ContinuousTransformerWrapper does not (`...layers.N.1.to_q`)."""
    base = f"{tower}.transformer"
    B, n, D = x.shape
    H = cfg.heads
    hd = D // H
    inv_freq = sd[f"{base}.attn_layers.rotary_pos_emb.inv_freq"]
    freqs = torch.arange(n).float()[:, None] * inv_freq[None, :]
    freqs = torch.cat((freqs, freqs), dim=-1)  # [n, 32]
    rd = freqs.shape[-1]
    cos, sin = freqs.cos(), freqs.sin()

    def rot(t):
        tl, tr = t[..., :rd], t[..., rd:]
        tl = tl * cos + _rotate_half(tl) * sin
        return torch.cat((tl, tr), dim=-1)

    for li in range(2 * cfg.depth):
        p = f"{base}.attn_layers.layers.{li}"
        h = _rmsnorm(x, sd[f"{p}.0.0.g"])
        if li % 2 == 0:
            q = (h @ sd[f"{p}.1{wrap}.to_q.weight"].t()).view(B, n, H, hd).transpose(1, 2)
            k = (h @ sd[f"{p}.1{wrap}.to_k.weight"].t()).view(B, n, H, hd).transpose(1, 2)
            v = (h @ sd[f"{p}.1{wrap}.to_v.weight"].t()).view(B, n, H, hd).transpose(1, 2)
            q, k, v = rot(q), rot(k), rot(v)
            att = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1)
            o = (att @ v).transpose(1, 2).reshape(B, n, D)
            o = o @ sd[f"{p}.1{wrap}.to_out.weight"].t() + sd[f"{p}.1{wrap}.to_out.bias"]
        else:
            u = h @ sd[f"{p}.1{wrap}.net.0.proj.weight"].t() + sd[f"{p}.1{wrap}.net.0.proj.bias"]
            a, gate = u.chunk(2, dim=-1)
            o = (a * F.gelu(gate)) @ sd[f"{p}.1{wrap}.net.3.weight"].t() + sd[f"{p}.1{wrap}.net.3.bias"]
        x = x + o
    return F.layer_norm(x, (D,), sd[f"{base}.norm.weight"], sd[f"{base}.norm.bias"], 1e-5)


def clvp_score(sd, cfg: CLVPConfig, text_tokens, codes):
    """CLVP.forward(return_loss=False) in eval mode (clvp.py:99-135).  text_tokens int [B, T],
    codes int64 [B, n] -> f32 [B]."""
    te = clvp_encoder(sd, cfg, "text_transformer", sd["text_emb.weight"][text_tokens.long()])
    se = clvp_encoder(sd, cfg, "speech_transformer", sd["speech_emb.weight"][codes.long()])
    tl = te.mean(dim=1) @ sd["to_text_latent.weight"].t()
    sl = se.mean(dim=1) @ sd["to_speech_latent.weight"].t()
    tl = F.normalize(tl, p=2, dim=-1)
    sl = F.normalize(sl, p=2, dim=-1)
    return (tl * sl).sum(-1) * sd["temperature"].exp()


# =============================================================================== CVVP
def cvvp_collapse(sd, cfg, tower, x):
    """CollapsingTransformer.forward in eval mode (cvvp.py:19-51): ContinuousTransformerWrapper(use_pos_emb=False) over the same
    x-transformers Encoder CLVP uses (ff_mult = 1) incl. its final LayerNorm (xtransformers.py:1187-1247), then pre_combiner =
    conv1x1 -> AttentionBlock (no relative positions) -> conv1x1 on [B, C, n], then the mean over time (the eval mask is all ones).
    x f32 [B, n, D] -> [B, out]."""
    h = clvp_encoder(sd, cfg, tower, x, wrap="").permute(0, 2, 1)
    h = F.conv1d(h, sd[f"{tower}.pre_combiner.0.weight"], sd[f"{tower}.pre_combiner.0.bias"])
    h = attention_block(sd, f"{tower}.pre_combiner.1", h, cfg.heads)
    h = F.conv1d(h, sd[f"{tower}.pre_combiner.2.weight"], sd[f"{tower}.pre_combiner.2.bias"])
    return h.mean(dim=-1)


def cvvp_forward(sd, cfg, mel_cond, codes):
    """CVVP.forward(mel_cond, mel_input, return_loss=False) in eval mode (cvvp.py:107-131) with mel_codes set (speech_emb is an
    embedding, cvvp.py:89-93).  mel_cond f32 [B, 80, T], codes int64 [B, n] -> f32 [B]."""
    c = F.conv1d(mel_cond.float(), sd["cond_emb.0.weight"], sd["cond_emb.0.bias"], stride=2, padding=2)
    c = F.conv1d(c, sd["cond_emb.1.weight"], sd["cond_emb.1.bias"], stride=2, padding=1).permute(0, 2, 1)
    cl = cvvp_collapse(sd, cfg, "conditioning_transformer", c) @ sd["to_conditioning_latent.weight"].t()
    sl = cvvp_collapse(sd, cfg, "speech_transformer", sd["speech_emb.emb.weight"][codes.long()]) @ sd["to_speech_latent.weight"].t()
    cl, sl = F.normalize(cl, p=2, dim=-1), F.normalize(sl, p=2, dim=-1)
    return (cl * sl).sum(-1) * sd["temperature"].exp()


def cvvp_score(sd, cfg, auto_conds, codes):
    """The CVVP term of the candidate ranking (api.py:464-468): the mean over the voice's conditioning clips of
    cvvp(clip repeated for every candidate, codes).  auto_conds f32 [1, n_clips, 80, T], codes int64 [B, n] -> f32 [B]."""
    B = codes.shape[0]
    acc = 0
    for cl in range(auto_conds.shape[1]):
        acc = acc + cvvp_forward(sd, cfg, auto_conds[:, cl].repeat(B, 1, 1), codes)
    return acc / auto_conds.shape[1]


def blend_candidate_scores(clvp_out, cvvp_out, cvvp_amount):
    """api.py:462-472: what the top-k runs on.  cvvp_amount == 1: CVVP alone; cvvp_out None (no conditioning clips, or amount 0): CLVP alone."""
    if cvvp_out is None or cvvp_amount == 0:
        return clvp_out
    if cvvp_amount == 1:
        return cvvp_out
    return cvvp_out * cvvp_amount + clvp_out * (1 - cvvp_amount)


# =============================================================================== diffusion network
def timestep_embedding(timesteps, dim, max_period=10000):
    """diffusion_decoder.py:21-39."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(x, w, b, groups=32):
    # GroupNorm32 (arch_util.py:21-41): 32 groups for every width used on the hot path, eps 1e-5
    return F.group_norm(x.float(), groups, w, b, 1e-5)


def rel_pos_bucket(rel, num_buckets=32, max_distance=64):
    """RelativePositionBias._relative_position_bucket, causal=False (xtransformers.py:155-175).
    rel = k_pos - q_pos (integer tensor)."""
    nb = num_buckets // 2
    n = -rel
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def rel_pos_bias(table, n, scale):
    """[H, n, n] additive bias = table[bucket(k - q)] * scale (xtransformers.py:177-186)."""
    pos = torch.arange(n)
    bucket = rel_pos_bucket(pos[None, :] - pos[:, None])
    return table[bucket].permute(2, 0, 1) * scale


def attention_block(sd, prefix, x, heads):
    """AttentionBlock + QKVAttentionLegacy with relative position bias (arch_util.py:80-123, 44-77).
    x [B, C, S].  qkv channels are laid out per head as [q(ch) k(ch) v(ch)] (reshape at line 63)."""
    B, C, S = x.shape
    ch = C // heads
    h = _gn(x, sd[f"{prefix}.norm.weight"], sd[f"{prefix}.norm.bias"])
    qkv = F.conv1d(h, sd[f"{prefix}.qkv.weight"], sd[f"{prefix}.qkv.bias"])
    q, k, v = qkv.reshape(B * heads, 3 * ch, S).split(ch, dim=1)
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    key = f"{prefix}.relative_pos_embeddings.relative_attention_bias.weight"
    if key in sd:
        w = (w.reshape(B, heads, S, S) + rel_pos_bias(sd[key], S, ch ** 0.5)[None]).reshape(B * heads, S, S)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(B, C, S)
    return x + F.conv1d(a, sd[f"{prefix}.proj_out.weight"], sd[f"{prefix}.proj_out.bias"])


def ar_get_conditioning(sd, cfg: ARConfig, mels):
    """UnifiedVoice.get_conditioning over ConditioningEncoder (autoregressive.py:204-228, 444-452): per clip
    conv1x1(80 -> D) -> 6 AttentionBlocks (no relative positions) -> time step 0; mean over clips.
    mels f32 [1, n_clips, 80, T] -> [1, D].  (conditioning front-end, oracle side.)"""
    conds = []
    for j in range(mels.shape[1]):
        h = F.conv1d(mels[:, j].float(), sd["conditioning_encoder.init.weight"], sd["conditioning_encoder.init.bias"])
        i = 0
        while f"conditioning_encoder.attn.{i}.norm.weight" in sd:
            h = attention_block(sd, f"conditioning_encoder.attn.{i}", h, cfg.heads)
            i += 1
        conds.append(h[:, :, 0])
    return torch.stack(conds, dim=1).mean(dim=1)


def diffusion_get_conditioning(sd, cfg: DiffusionConfig, mels):
    """DiffusionTts.get_conditioning over contextual_embedder (diffusion_decoder.py:186-192, 222-230): per clip
    conv k3 stride 2 (100 -> C), conv k3 stride 2 (C -> 2C), 5 AttentionBlocks with relative positions; the clips are
    concatenated along time and averaged.  mels f32 [1, n_clips, 100, T] -> [1, 2C]."""
    outs = []
    for j in range(mels.shape[1]):
        h = F.conv1d(mels[:, j].float(), sd["contextual_embedder.0.weight"], sd["contextual_embedder.0.bias"], stride=2, padding=1)
        h = F.conv1d(h, sd["contextual_embedder.1.weight"], sd["contextual_embedder.1.bias"], stride=2, padding=1)
        i = 2
        while f"contextual_embedder.{i}.norm.weight" in sd:
            h = attention_block(sd, f"contextual_embedder.{i}", h, cfg.num_heads)
            i += 1
        outs.append(h)
    return torch.cat(outs, dim=-1).mean(dim=-1)


def res_block(sd, prefix, x, emb):
    """ResBlock(use_scale_shift_norm=True, efficient_config=True, kernel 3) (diffusion_decoder.py:60-120)."""
    h = F.silu(_gn(x, sd[f"{prefix}.in_layers.0.weight"], sd[f"{prefix}.in_layers.0.bias"]))
    h = F.conv1d(h, sd[f"{prefix}.in_layers.2.weight"], sd[f"{prefix}.in_layers.2.bias"])
    e = F.linear(F.silu(emb), sd[f"{prefix}.emb_layers.1.weight"], sd[f"{prefix}.emb_layers.1.bias"])
    scale, shift = e[..., None].chunk(2, dim=1)
    h = _gn(h, sd[f"{prefix}.out_layers.0.weight"], sd[f"{prefix}.out_layers.0.bias"]) * (1 + scale) + shift
    h = F.conv1d(F.silu(h), sd[f"{prefix}.out_layers.3.weight"], sd[f"{prefix}.out_layers.3.bias"], padding=1)
    return x + h


def diffusion_layer(sd, prefix, x, emb, heads):
    # DiffusionLayer (diffusion_decoder.py:123-131)
    return attention_block(sd, f"{prefix}.attn", res_block(sd, f"{prefix}.resblk", x, emb), heads)


def diffusion_timestep_independent(sd, cfg: DiffusionConfig, latents, cond_latent, expected_seq_len):
    """DiffusionTts.timestep_independent, latent branch, eval (diffusion_decoder.py:232-255).
    latents [B, M, 1024]; cond_latent [B, 2C] -> [B, C, S]."""
    x = latents.permute(0, 2, 1)
    cond_scale, cond_shift = cond_latent.chunk(2, dim=1)
    h = F.conv1d(x, sd["latent_conditioner.0.weight"], sd["latent_conditioner.0.bias"], padding=1)
    for i in range(1, 5):
        h = attention_block(sd, f"latent_conditioner.{i}", h, cfg.num_heads)
    h = _gn(h, sd["code_norm.weight"], sd["code_norm.bias"]) * (1 + cond_scale[..., None]) + cond_shift[..., None]
    return F.interpolate(h, size=expected_seq_len, mode="nearest")


def diffusion_forward(sd, cfg: DiffusionConfig, x, timesteps, code_emb, conditioning_free=False):
    """DiffusionTts.forward with precomputed_aligned_embeddings (diffusion_decoder.py:262-322).
    The `extraneous_addition * 0` term (314-318) is exactly zero for finite weights and is dropped."""
    C = cfg.model_channels
    if conditioning_free:
        code_emb = sd["unconditioned_embedding"].repeat(x.shape[0], 1, x.shape[-1])
    t = timestep_embedding(timesteps, C)
    t = F.linear(F.silu(F.linear(t, sd["time_embed.0.weight"], sd["time_embed.0.bias"])),
                 sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    for i in range(3):
        code_emb = diffusion_layer(sd, f"conditioning_timestep_integrator.{i}", code_emb, t, cfg.num_heads)
    h = F.conv1d(x, sd["inp_block.weight"], sd["inp_block.bias"], padding=1)
    h = F.conv1d(torch.cat([h, code_emb], dim=1), sd["integrating_conv.weight"], sd["integrating_conv.bias"])
    for i in range(cfg.num_layers):
        h = diffusion_layer(sd, f"layers.{i}", h, t, cfg.num_heads)
    for i in range(cfg.num_layers, cfg.num_layers + 3):
        h = res_block(sd, f"layers.{i}", h, t)
    h = F.silu(_gn(h.float(), sd["out.0.weight"], sd["out.0.bias"]))
    return F.conv1d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


# =============================================================================== diffusion sampler
def space_timesteps(num_timesteps, count):
    """utils/diffusion.py:1152-1205 for a single section (api.py:68 passes [desired_steps])."""
    if count <= 1:
        stride = 1
    else:
        stride = (num_timesteps - 1) / (count - 1)
    cur = 0.0
    taken = []
    for _ in range(count):
        taken.append(round(cur))
        cur += stride
    return sorted(set(taken))


class Schedule:
    """SpacedDiffusion(linear betas over 4000 steps, learned_range, epsilon) tables in float64
    (utils/diffusion.py:94-111, 192-249, 1102-1116), extracted to f32 per lookup like
    _extract_into_tensor (1237-1250)."""

    def __init__(self, steps, trained_steps=4000, cond_free=True, cond_free_k=2.0):
        scale = 1000 / trained_steps
        base_betas = np.linspace(scale * 0.0001, scale * 0.02, trained_steps, dtype=np.float64)
        base_ac = np.cumprod(1.0 - base_betas, axis=0)
        use = set(space_timesteps(trained_steps, steps))
        last = 1.0
        betas, tmap = [], []
        for i, ac in enumerate(base_ac):
            if i in use:
                betas.append(1 - ac / last)
                last = ac
                tmap.append(i)
        betas = np.array(betas, dtype=np.float64)
        self.timestep_map = np.array(tmap, dtype=np.int64)
        self.num_timesteps = len(betas)
        self.cond_free = cond_free
        self.cond_free_k = cond_free_k
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.betas = betas
        self.sqrt_recip_ac = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_ac = np.sqrt(1.0 / ac - 1)
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.post_logvar_clipped = np.log(np.append(post_var[1], post_var[1:]))
        self.log_betas = np.log(betas)
        self.coef1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
        self.coef2 = (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)

    def f32(self, arr, i):
        return float(np.float32(arr[i]))


def p_sample_step(sched: Schedule, x, i, out_cond, out_uncond, noise):
    """GaussianDiffusion.p_mean_variance + p_sample for spaced index i (utils/diffusion.py:312-418,
    487-531).  out_* are raw model outputs [B, 2C, S]; noise is the randn_like(x) draw."""
    C = x.shape[1]
    eps, var_values = out_cond[:, :C], out_cond[:, C:]
    min_log = sched.f32(sched.post_logvar_clipped, i)
    max_log = sched.f32(sched.log_betas, i)
    frac = (var_values + 1) / 2
    log_var = frac * max_log + (1 - frac) * min_log
    if sched.cond_free:
        cfk = sched.cond_free_k * (1 - i / sched.num_timesteps)
        eps = (1 + cfk) * eps - cfk * out_uncond[:, :C]
    x0 = (sched.f32(sched.sqrt_recip_ac, i) * x - sched.f32(sched.sqrt_recipm1_ac, i) * eps).clamp(-1, 1)
    mean = sched.f32(sched.coef1, i) * x0 + sched.f32(sched.coef2, i) * x
    nonzero = 0.0 if i == 0 else 1.0
    return mean + nonzero * torch.exp(0.5 * log_var) * noise


def p_sample_loop(sd, cfg: DiffusionConfig, sched: Schedule, code_emb, x_T, step_noise):
    """SpacedDiffusion.p_sample_loop (utils/diffusion.py:533-621; _WrappedModel 1215-1220).
    step_noise [N, B, C, S]: step_noise[i] is the draw used at spaced index i."""
    x = x_T
    B = x.shape[0]
    for i in reversed(range(sched.num_timesteps)):
        ts = torch.full((B,), int(sched.timestep_map[i]), dtype=torch.long)
        oc = diffusion_forward(sd, cfg, x, ts, code_emb, False)
        ou = diffusion_forward(sd, cfg, x, ts, code_emb, True) if sched.cond_free else None
        x = p_sample_step(sched, x, i, oc, ou, step_noise[i])
    return x


def denormalize_tacotron_mel(m):
    # utils/audio.py:59-64
    return ((m + 1) / 2) * (TACOTRON_MEL_MAX - TACOTRON_MEL_MIN) + TACOTRON_MEL_MIN


def do_spectrogram_diffusion(sd, cfg, sched, latents, cond_latent, x_T, step_noise):
    """api.py:117-130 with the random draws injected."""
    S = latents.shape[1] * 4 * 24000 // 22050
    code_emb = diffusion_timestep_independent(sd, cfg, latents, cond_latent, S)
    mel = p_sample_loop(sd, cfg, sched, code_emb, x_T, step_noise)
    return denormalize_tacotron_mel(mel)[:, :, :S]


# =============================================================================== UnivNet
def _lrelu(x, s):
    return F.leaky_relu(x, s)


def kernel_predictor(sd, cfg: VocoderConfig, p, c):
    """KernelPredictor.forward (vocoder.py:66-93).  sd holds *folded* weights."""
    s = cfg.lrelu_slope
    h = _lrelu(F.conv1d(c, sd[f"{p}.input_conv.0.weight"], sd[f"{p}.input_conv.0.bias"], padding=2), s)
    for r in range(3):
        t = _lrelu(F.conv1d(h, sd[f"{p}.residual_convs.{r}.1.weight"], sd[f"{p}.residual_convs.{r}.1.bias"], padding=1), s)
        t = _lrelu(F.conv1d(t, sd[f"{p}.residual_convs.{r}.3.weight"], sd[f"{p}.residual_convs.{r}.3.bias"], padding=1), s)
        h = h + t
    k = F.conv1d(h, sd[f"{p}.kernel_conv.weight"], sd[f"{p}.kernel_conv.bias"], padding=1)
    b = F.conv1d(h, sd[f"{p}.bias_conv.weight"], sd[f"{p}.bias_conv.bias"], padding=1)
    B, _, L = c.shape
    ch = cfg.channel_size
    nl = len(cfg.dilations)
    return k.view(B, nl, ch, 2 * ch, 3, L), b.view(B, nl, 2 * ch, L)


def location_variable_convolution(x, kernel, bias, hop):
    """vocoder.py:182-216 with dilation=1: out[b,o,l*hop+s] = bias[b,o,l] +
    sum_{i,k} xpad[b,i,l*hop+s+k] * kernel[b,i,o,k,l]."""
    B, Cin, T = x.shape
    L = kernel.shape[-1]
    assert T == L * hop
    xp = F.pad(x, (1, 1))
    win = xp.unfold(2, hop + 2, hop)        # [B, Cin, L, hop+2]
    win = win.unfold(3, 3, 1)               # [B, Cin, L, hop, 3]
    o = torch.einsum("bilsk,biokl->bols", win, kernel) + bias[..., None]
    return o.reshape(B, -1, T)


def univnet_forward(sd, cfg: VocoderConfig, c, z):
    """UnivNetGenerator.forward (vocoder.py:267-282) + LVCBlock.forward (155-180)."""
    s = cfg.lrelu_slope
    ch = cfg.channel_size
    x = F.conv1d(F.pad(z, (3, 3), mode="reflect"), sd["conv_pre.weight"], sd["conv_pre.bias"])
    hop = 1
    for bi, stride in enumerate(cfg.strides):
        hop *= stride
        p = f"res_stack.{bi}"
        x = F.conv_transpose1d(_lrelu(x, s), sd[f"{p}.convt_pre.1.weight"], sd[f"{p}.convt_pre.1.bias"],
                               stride=stride, padding=stride // 2 + stride % 2, output_padding=stride % 2)
        kernels, bias = kernel_predictor(sd, cfg, f"{p}.kernel_predictor", c)
        for j, dil in enumerate(cfg.dilations):
            o = F.conv1d(_lrelu(x, s), sd[f"{p}.conv_blocks.{j}.1.weight"], sd[f"{p}.conv_blocks.{j}.1.bias"],
                         padding=dil, dilation=dil)
            o = _lrelu(o, s)
            o = location_variable_convolution(o, kernels[:, j], bias[:, j], hop)
            x = x + torch.sigmoid(o[:, :ch]) * torch.tanh(o[:, ch:])
    x = _lrelu(x, s)
    x = F.conv1d(F.pad(x, (3, 3), mode="reflect"), sd["conv_post.1.weight"], sd["conv_post.1.bias"])
    return torch.tanh(x)


def univnet_inference(sd, cfg: VocoderConfig, mel, z):
    """UnivNetGenerator.inference (vocoder.py:300-312): 10 pad frames of -11.5129, z injected
    ([B, 64, S+10]), drop the last 10*hop samples, clamp."""
    pad = torch.full((mel.shape[0], cfg.n_mel_channels, 10), -11.5129)
    audio = univnet_forward(sd, cfg, torch.cat((mel, pad), dim=2), z)
    return audio[:, :, :-(cfg.hop_length * 10)].clamp(-1, 1)


def hifigan_inference(sd, cfg, latents, g):
    """HifiganGenerator.inference + forward (hifigan_decoder.py:229-289) on weight-norm-folded weights:
    latents [B, T, in_channels] -> linear x4 -> linear x24000/22050 -> conv_pre + cond_layer(g) -> per stage
    [lrelu(0.1) -> ConvTranspose1d -> mean of the ResBlock1 stack] -> lrelu(0.01) -> conv_post -> tanh.
    g [B, cond_channels] (the AR conditioning latent); returns [B, 1, T2 * hop]."""
    up1 = F.interpolate(latents.float().transpose(1, 2), scale_factor=[1024 / 256], mode="linear")
    x = F.interpolate(up1, scale_factor=[24000 / 22050], mode="linear")
    o = F.conv1d(x, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    o = o + F.conv1d(g.float().unsqueeze(0).transpose(1, 2), sd["cond_layer.weight"], sd["cond_layer.bias"])
    nk = len(cfg.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(cfg.upsample_factors, cfg.upsample_kernel_sizes)):
        o = F.leaky_relu(o, cfg.lrelu_slope)
        o = F.conv_transpose1d(o, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        z_sum = None
        for j, ks in enumerate(cfg.resblock_kernel_sizes):
            xr = o
            p = f"resblocks.{i * nk + j}"
            for dd, dil in enumerate(cfg.resblock_dilation_sizes):  # ResBlock1.forward (hifigan_decoder.py:90-107)
                xt = F.leaky_relu(xr, cfg.lrelu_slope)
                xt = F.conv1d(xt, sd[f"{p}.convs1.{dd}.weight"], sd[f"{p}.convs1.{dd}.bias"], dilation=dil, padding=(ks * dil - dil) // 2)
                xt = F.leaky_relu(xt, cfg.lrelu_slope)
                xt = F.conv1d(xt, sd[f"{p}.convs2.{dd}.weight"], sd[f"{p}.convs2.{dd}.bias"], padding=(ks - 1) // 2)
                xr = xt + xr
            z_sum = xr if z_sum is None else z_sum + xr
        o = z_sum / nk
    o = F.leaky_relu(o)  # default slope 0.01 (hifigan_decoder.py:257)
    o = F.conv1d(o, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(o)

