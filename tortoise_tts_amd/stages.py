"""Host-side stage objects: one per C-ABI handle.  They own the packed weights, translate between
the reference's tensor conventions (channels-first, int64 ids) and the engine's, and keep every
call on the caller's current torch stream.  No model arithmetic happens here beyond embedding
gathers for the once-per-utterance prefix (plumbing).
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import engine as E
from . import pack
from .config import ARConfig, CLVPConfig, CVVPConfig, DiffusionConfig, VocoderConfig
from .schedule import Schedule


def _i32(t, device):
    return t.to(device=device, dtype=torch.int32).contiguous()


def latent_pass_embeddings(text_emb_w, text_pos_w, mel_emb_w, mel_pos_w, cfg, cond, text_tokens, codes, stream_positions=False):
    """Input rows of the teacher-forced pass (autoregressive.py:454-506): [cond | start text stop | start codes stop] with their
    learned positions -> (emb f32 [k, 1 + T + 2 + n + 2, D], number of mel rows).  Plain tensor indexing (device or host)."""
    k, n = codes.shape
    t = F.pad(text_tokens.long(), (0, 1), value=cfg.stop_text_token)
    t = F.pad(t, (1, 0), value=cfg.start_text_token)
    text_emb = text_emb_w[t] + text_pos_w[: t.shape[1]][None]
    m = F.pad(codes.long(), (0, 1), value=cfg.stop_mel_token)
    m = F.pad(m, (1, 0), value=cfg.start_mel_token)
    pos = torch.arange(m.shape[1], device=m.device)
    if stream_positions:
        pos = torch.where(pos > 0, pos + 1, pos)
    mel_emb = mel_emb_w[m] + mel_pos_w[pos][None]
    if text_emb.shape[0] == 1 and k > 1:
        text_emb = text_emb.expand(k, -1, -1)
    if cond.shape[0] == 1 and k > 1:
        cond = cond.expand(k, -1)
    return torch.cat([cond[:, None, :], text_emb, mel_emb], dim=1).contiguous(), m.shape[1]


class ArStage:
    """UnifiedVoice hot path: prefill + sampling loop + latent re-pass (autoregressive.py:454-563)."""

    def __init__(self, sd, cfg: ARConfig = ARConfig(), device="cuda", dtype=E.TT_BF16, max_batch=256, max_text=402,
                 max_new_tokens=500, max_latent_candidates=4, share_weights_with=None, kv_cache=True, max_groups=1):
        self.lib = E.init()
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = dtype
        # several handles (one per concurrent decode stream) can share one packed copy of the weights
        self.w = share_weights_with.w if share_weights_with is not None else pack.pack_ar(sd, cfg, self.device, dtype)
        c = E.ArConfig()
        c.dtype = dtype
        c.layers, c.model_dim, c.heads = cfg.layers, cfg.model_dim, cfg.heads
        c.vocab = cfg.number_mel_codes
        c.start_mel_token, c.stop_mel_token = cfg.start_mel_token, cfg.stop_mel_token
        c.mel_pos_len = cfg.mel_pos_len
        c.max_batch = max_batch
        c.max_prefix = 1 + max_text + 2 + 1
        c.max_new_tokens = max_new_tokens + 2
        c.max_full_rows = max_latent_candidates * (1 + max_text + 2 + max_new_tokens + 2)
        # TextToSpeech(kv_cache=...) only changes WHICH mel position row a generated token gets (autoregressive.py:134-149):
        # the engine always keeps a KV cache; kv_cache=False (the reference default) selects rows 0,1,2,... instead of 0,2,3,...
        c.mel_pos_offset = 2 if kv_cache else 1
        c.max_groups = max_groups  # utterances decoded in one batch (prefill_group); max_batch counts the sequences of all of them
        self.max_groups = max_groups
        self.max_latent_candidates = max_latent_candidates
        self.ccfg = c
        self.h = E.vp()
        E.check(self.lib.tt_ar_create(C.byref(c), C.byref(self.w.weights), C.byref(self.h)))
        # A/B switches of the measurement scripts (scripts/ab_stage.py); the product default is what tt_ar_create sets
        for env, opt in (("TT_AR_LOOKAHEAD", E.TT_AR_OPT_LOOKAHEAD),):
            if os.environ.get(env):
                self.set_option(opt, int(os.environ[env]))

    def set_option(self, option, value):
        """tt_ar_set_option: host lookahead of the paced decode loop (no effect on the codes)."""
        E.check(self.lib.tt_ar_set_option(self.h, int(option), int(value)))

    def stat(self, which):
        """tt_ar_stat: 0 decode-step graph captures, 1 queue drains of the launch loop, 2 launches per decode step."""
        return self.lib.tt_ar_stat(self.h, int(which))

    def guard(self, reset=True):
        """Non-finite values met by this stage's norms / sampler since the last reset (operand-overflow guard)."""
        return E.guard_count(self.lib.tt_ar_guard(self.h, int(reset)))

    def close(self):
        if self.h:
            self.lib.tt_ar_destroy(self.h)
            self.h = E.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- prefix (autoregressive.py:538-544)
    def prefix_embedding(self, cond_latent, text_tokens):
        cfg = self.cfg
        t = F.pad(text_tokens.to(self.device).long(), (0, 1), value=cfg.stop_text_token)
        t = F.pad(t, (1, 0), value=cfg.start_text_token)
        n = t.shape[1]
        text_emb = self.w.text_emb[t] + self.w.text_pos[:n][None]
        return torch.cat([cond_latent.to(self.device).float()[:, None, :], text_emb], dim=1)  # [1, P, D]

    def prefill(self, cond_latent, text_tokens):
        emb = self.prefix_embedding(cond_latent[:1], text_tokens[:1])[0].contiguous()
        self.P = emb.shape[0]
        E.check(self.lib.tt_ar_prefill(self.h, E.ptr(emb), self.P, E.stream_ptr()))

    def prefill_group(self, group, n_groups, cond_latent, text_tokens):
        """Prefix of utterance `group` of a batch of n_groups utterances that generate() then decodes together (sequences
        [group * B / n_groups, (group + 1) * B / n_groups)).  Call for group 0 first."""
        emb = self.prefix_embedding(cond_latent[:1], text_tokens[:1])[0].contiguous()
        self.P = emb.shape[0]
        E.check(self.lib.tt_ar_prefill_group(self.h, group, n_groups, E.ptr(emb), self.P, E.stream_ptr()))

    def logits(self, rows):
        out = torch.empty(rows, self.cfg.number_mel_codes, device=self.device, dtype=torch.float32)
        E.check(self.lib.tt_ar_get_logits(self.h, E.ptr(out), rows, E.stream_ptr()))
        return out

    def begin(self, B):
        E.check(self.lib.tt_ar_begin(self.h, B, E.stream_ptr()))

    def decode_step(self, tokens):
        t = _i32(tokens, self.device)
        E.check(self.lib.tt_ar_decode_step(self.h, E.ptr(t), E.stream_ptr()))

    def _codes_buffer(self, B, max_new):
        """int32 [B, max_new] receiving buffer of one call (the sampler itself writes a buffer the handle owns, so this address is
        not part of the kept decode-step graph's key)."""
        return torch.empty(B, max_new, device=self.device, dtype=torch.int32)

    def generate(self, B, max_new, temperature=0.8, top_p=0.8, repetition_penalty=2.0, top_k=50, seed=0, row_offset=0,
                 exp_noise=None, group_seeds=None, typical_mass=0.0):
        """Returns (codes int64 [B, n_steps], n_steps).  exp_noise: optional f32 [max_new, B, V] Exp(1) draws.
        group_seeds: after prefill_group calls, one Philox key per utterance (default: `seed` for all of them).
        typical_mass: 0 < mass < 1 = tts(typical_sampling=True, typical_mass=mass) (autoregressive.py:558); 0 = off."""
        s = E.Sampling()
        s.temperature, s.top_p, s.repetition_penalty, s.top_k = temperature, top_p, repetition_penalty, top_k
        s.typical_mass = float(typical_mass)
        s.seed, s.row_offset = seed, row_offset
        if group_seeds is not None:
            gs = (C.c_ulonglong * len(group_seeds))(*[int(v) for v in group_seeds])
            s.group_seeds = C.cast(gs, C.POINTER(C.c_ulonglong))
        if exp_noise is not None:
            exp_noise = exp_noise.to(device=self.device, dtype=torch.float32).contiguous()
            assert exp_noise.shape == (max_new, B, self.cfg.number_mel_codes)
        s.exp_noise = E.ptr(exp_noise)
        codes = self._codes_buffer(B, max_new)
        n = C.c_int(0)
        E.check(self.lib.tt_ar_generate(self.h, B, max_new, C.byref(s), E.ptr(codes), C.byref(n), E.stream_ptr()))
        return codes[:, :n.value].long(), n.value

    def generate_stream(self, B, max_new, chunk, first_chunk=None, temperature=0.8, top_p=0.8, repetition_penalty=2.0, top_k=50, seed=0,
                        row_offset=0, typical_mass=0.0):
        """Generator over the sampling loop in pieces (api_fast.py:389-420 pulls get_generator() token by token and decodes every
        `stream_chunk_size` tokens): yields (codes int64 [B, n_so_far], finished) after each chunk."""
        s = E.Sampling()
        s.temperature, s.top_p, s.repetition_penalty, s.top_k = temperature, top_p, repetition_penalty, top_k
        s.seed, s.row_offset = seed, row_offset
        s.typical_mass = float(typical_mass)
        s.exp_noise = None
        codes = self._codes_buffer(B, max_new)
        n, fin = C.c_int(0), C.c_int(0)
        first = True
        while True:
            want = min((first_chunk or chunk) if first else chunk, max_new - n.value)
            if want <= 0:
                return
            E.check(self.lib.tt_ar_generate_chunk(self.h, B, 1 if first else 0, want, max_new, C.byref(s), E.ptr(codes), C.byref(n),
                                                  C.byref(fin), E.stream_ptr()))
            first = False
            done = bool(fin.value) or n.value >= max_new
            yield codes[:, :n.value].long(), done
            if done:
                return

    def stream_latents(self, B, n):
        """f32 [B, n, D]: the per-step latents the decode loop filed for the first n tokens of the running generation - the latent
        half of the (token, latent) pairs of the reference's streaming generator (api_fast.py:402-411); max_batch <= 8 handles."""
        out = torch.empty(B, n, self.cfg.model_dim, device=self.device, dtype=torch.float32)
        E.check(self.lib.tt_ar_stream_latents(self.h, B, n, E.ptr(out), E.stream_ptr()))
        return out

    # -- latent re-pass (autoregressive.py:454-506 as api.py:521-524 calls it)
    def latents(self, cond_latent, text_tokens, codes, stream_positions=False):
        """stream_positions: mel positions 0, 2, 3, ... instead of 0, 1, 2, ... - the per-step states the reference's streaming
        path collects under kv_cache=True (api_fast.py:389-414 with the cached-decode position rule of autoregressive.py:134-149)."""
        emb, mel_rows = latent_pass_embeddings(self.w.text_emb, self.w.text_pos, self.w.mel_emb, self.w.mel_pos, self.cfg,
                                               cond_latent.to(self.device).float(), text_tokens.to(self.device), codes.to(self.device),
                                               stream_positions)
        k = codes.shape[0]
        outs = []
        for i in range(0, k, self.max_latent_candidates):
            e = emb[i:i + self.max_latent_candidates].contiguous()
            out = torch.empty_like(e)
            E.check(self.lib.tt_ar_latents(self.h, E.ptr(e), e.shape[0], e.shape[1], E.ptr(out), E.stream_ptr()))
            outs.append(out)
        out = torch.cat(outs, dim=0)
        # enc = hidden[:, 1:]; mel part = last mel_rows rows; drop the final two (autoregressive.py:425-431, 503)
        return out[:, -mel_rows:][:, :-2]


class ClvpStage:
    """CLVP.forward(return_loss=False) (clvp.py:99-135)."""

    def __init__(self, sd, cfg: CLVPConfig = CLVPConfig(), device="cuda", dtype=E.TT_BF16, max_rows=256 * 500):
        self.lib = E.init()
        self.cfg = cfg
        self.device = torch.device(device)
        self.w = pack.pack_clvp(sd, cfg, self.device, dtype)
        c = E.ClvpConfig()
        c.dtype, c.dim, c.latent_dim, c.depth, c.heads = dtype, cfg.dim, cfg.dim_latent, cfg.depth, cfg.heads
        c.ff_inner, c.rot_dim, c.max_rows = cfg.dim * cfg.ff_mult, cfg.rotary_dim, max_rows
        self.max_rows = max_rows
        self.h = E.vp()
        E.check(self.lib.tt_clvp_create(C.byref(c), C.byref(self.w.text), C.byref(self.w.speech), E.ptr(self.w.temperature),
                                        C.byref(self.h)))

    def close(self):
        if self.h:
            self.lib.tt_clvp_destroy(self.h)
            self.h = E.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def guard(self, reset=True):
        return E.guard_count(self.lib.tt_clvp_guard(self.h, int(reset)))

    def score(self, text_tokens, codes):
        """text_tokens int [1 or B, T] (rows identical), codes int [B, n] -> f32 [B]."""
        text = _i32(text_tokens[0], self.device)
        B, n = codes.shape
        outs = []
        per = max(1, self.max_rows // n)
        for i in range(0, B, per):
            c = _i32(codes[i:i + per], self.device)
            out = torch.empty(c.shape[0], device=self.device, dtype=torch.float32)
            E.check(self.lib.tt_clvp_score(self.h, E.ptr(text), text.shape[0], E.ptr(c), c.shape[0], n, E.ptr(out), E.stream_ptr()))
            outs.append(out)
        return torch.cat(outs)

    def score_groups(self, texts, codes):
        """Several utterances of one voice (long-form reading): texts = list of G int tensors [1, T_g], codes int [G * N, n] with the N
        candidates of utterance g in rows [g * N, (g + 1) * N) -> f32 [G * N].  ONE speech-tower pass for as many utterances as the
        handle's capacity holds (at most 16 per pass); every score equals score() on that utterance alone, bit for bit."""
        G = len(texts)
        GN, n = codes.shape
        assert G >= 1 and GN % G == 0
        N = GN // G
        per = max(1, min(16, self.max_rows // max(1, N * n)))
        if N * n > self.max_rows:  # one utterance's candidates do not fit one pass: fall back to the chunked single-utterance form
            return torch.cat([self.score(texts[g], codes[g * N:(g + 1) * N]) for g in range(G)])
        outs = []
        for g0 in range(0, G, per):
            sel = texts[g0:g0 + per]
            flat = torch.cat([_i32(t.reshape(-1), self.device) for t in sel])
            lens = (C.c_int * len(sel))(*[int(t.numel()) for t in sel])
            c = _i32(codes[g0 * N:(g0 + len(sel)) * N], self.device)
            out = torch.empty(c.shape[0], device=self.device, dtype=torch.float32)
            E.check(self.lib.tt_clvp_score_groups(self.h, E.ptr(flat), lens, len(sel), E.ptr(c), N, n, E.ptr(out), E.stream_ptr()))
            outs.append(out)
        return torch.cat(outs)


def nearest_interp_index(m, s):
    """Source row of F.interpolate(mode='nearest') for each of s outputs given m inputs
    (ATen nearest_neighbor_compute_source_index: floor(dst * float(m / s)), clamped)."""
    scale = np.float32(m) / np.float32(s)
    idx = np.floor(np.arange(s, dtype=np.float32) * scale).astype(np.int64)
    return np.minimum(idx, m - 1).astype(np.int32)


class CvvpStage:
    """The CVVP term of the candidate ranking, tts(cvvp_amount > 0) (cvvp.py:107-131 as api.py:464-468 drives it)."""

    def __init__(self, sd, cfg: CVVPConfig = CVVPConfig(), device="cuda", dtype=E.TT_F16, max_rows=256 * 500, max_cond_frames=520):
        self.lib = E.init()
        self.cfg = cfg
        self.device = torch.device(device)
        self.w = pack.pack_cvvp(sd, cfg, self.device, dtype)
        c = E.CvvpConfig()
        c.dtype, c.dim, c.heads, c.depth, c.rot_dim = dtype, cfg.model_dim, cfg.heads, cfg.depth, cfg.rotary_dim
        c.mel_channels, c.mel_pad, c.max_rows, c.max_cond_frames = cfg.mel_channels, self.w.mel_pad, max_rows, max_cond_frames
        self.max_rows, self.max_cond_frames = max_rows, max_cond_frames
        self.h = E.vp()
        E.check(self.lib.tt_cvvp_create(C.byref(c), C.byref(self.w.weights), C.byref(self.h)))

    def close(self):
        if self.h:
            self.lib.tt_cvvp_destroy(self.h)
            self.h = E.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def guard(self, reset=True):
        return E.guard_count(self.lib.tt_cvvp_guard(self.h, int(reset)))

    def score(self, auto_conds, codes):
        """auto_conds f32 [1, n_clips, 80, T] (the voice's conditioning clips, api.py:262-276), codes int [B, n] -> f32 [B]: the mean over
        the clips of cvvp(clip, codes) (api.py:464-468)."""
        mels = auto_conds.to(self.device).float().reshape(-1, auto_conds.shape[-2], auto_conds.shape[-1]).contiguous()
        n_clips, _, T = mels.shape
        if T > self.max_cond_frames:
            raise ValueError(f"conditioning clips of {T} mel frames exceed this CVVP handle's capacity ({self.max_cond_frames})")
        B, n = codes.shape
        outs = []
        per = max(1, self.max_rows // n)
        for i in range(0, B, per):
            c = _i32(codes[i:i + per], self.device)
            out = torch.empty(c.shape[0], device=self.device, dtype=torch.float32)
            E.check(self.lib.tt_cvvp_score(self.h, E.ptr(mels), n_clips, T, E.ptr(c), c.shape[0], n, E.ptr(out), E.stream_ptr()))
            outs.append(out)
        return torch.cat(outs)


class DiffusionStage:
    """DiffusionTts + SpacedDiffusion.p_sample_loop (api.py:117-130)."""

    def __init__(self, sd, cfg: DiffusionConfig = DiffusionConfig(), device="cuda", dtype=E.TT_BF16, max_seq=2304, max_codes=512,
                 max_steps=512, max_batch=1):
        self.lib = E.init()
        self.cfg = cfg
        self.device = torch.device(device)
        self.w = pack.pack_diffusion(sd, cfg, self.device, dtype)
        c = E.DiffConfig()
        c.dtype, c.channels, c.heads, c.num_layers = dtype, cfg.model_channels, cfg.num_heads, cfg.num_layers
        c.in_channels, c.in_pad, c.out_channels = cfg.in_channels, self.w.in_pad, cfg.out_channels
        c.latent_channels, c.max_seq, c.max_codes, c.max_steps = cfg.in_latent_channels, max_seq, max_codes, max_steps
        c.max_batch = max_batch  # utterances one sample_many() pass may hold
        self.max_batch = max_batch
        self.h = E.vp()
        E.check(self.lib.tt_diff_create(C.byref(c), C.byref(self.w.weights), C.byref(self.h)))
        self.S = 0
        if os.environ.get("TT_DIFF_OVERLAP_PREPASS"):  # A/B switch of the measurement scripts
            self.set_option(E.TT_DIFF_OPT_OVERLAP_PREPASS, int(os.environ["TT_DIFF_OVERLAP_PREPASS"]))
        if os.environ.get("TT_DIFF_FUSED_GN"):
            self.set_option(E.TT_DIFF_OPT_FUSED_GN, int(os.environ["TT_DIFF_FUSED_GN"]))

    def set_option(self, option, value):
        E.check(self.lib.tt_diff_set_option(self.h, int(option), int(value)))

    def close(self):
        if self.h:
            self.lib.tt_diff_destroy(self.h)
            self.h = E.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stat(self, which):
        return self.lib.tt_diff_stat(self.h, int(which))

    def guard(self, reset=True):
        """Non-finite GroupNorm statistics / sampler inputs since the last reset, as of the last finished sampling run (the caller
        has synchronised, e.g. by reading the result)."""
        return E.guard_count(self.lib.tt_diff_guard(self.h, int(reset)))

    def condition(self, latents, cond_latent, S):
        """latents f32 [1, M, latent]; cond_latent f32 [1, 2C] (diffusion_decoder.py:232-260)."""
        lat = latents[0].to(self.device).float().contiguous()
        cond = cond_latent[0].to(self.device).float().contiguous()
        idx = torch.from_numpy(nearest_interp_index(lat.shape[0], S)).to(self.device)
        E.check(self.lib.tt_diff_condition(self.h, E.ptr(lat), lat.shape[0], E.ptr(cond), E.ptr(idx), S, E.stream_ptr()))
        self.S = S

    def code_emb(self):
        out = torch.empty(self.S, self.cfg.model_channels, device=self.device, dtype=torch.float32)
        E.check(self.lib.tt_diff_get_code_emb(self.h, E.ptr(out), E.stream_ptr()))
        return out.t()[None]  # [1, C, S] like the reference

    def forward(self, x, timestep, cond_free=True):
        """x f32 [1, 100, S] -> raw model outputs [B, 200, S] (B = 2 with cond_free: row 0 cond, row 1 uncond)."""
        if self.S <= 0:
            raise ValueError("forward() needs condition() first (after sample_many the handle holds a batch)")
        xt = x[0].to(self.device).float().t().contiguous()
        B = 2 if cond_free else 1
        out = torch.empty(B, self.S, self.cfg.out_channels, device=self.device, dtype=torch.float32)
        E.check(self.lib.tt_diff_forward(self.h, E.ptr(xt), int(timestep), int(cond_free), E.ptr(out), E.stream_ptr()))
        return out.permute(0, 2, 1)

    @staticmethod
    def _run_order_steps(sched: Schedule):
        """tt_diff_step records in the order the steps run (i = N-1 ... 0) + that order."""
        N = sched.num_timesteps
        steps = (E.DiffStep * N)()
        order = list(reversed(range(N)))
        for j, i in enumerate(order):
            st = steps[j]
            st.timestep = int(sched.timestep_map[i])
            st.min_log = sched.f32(sched.post_logvar_clipped, i)
            st.max_log = sched.f32(sched.log_betas, i)
            st.cfk = float(np.float32(sched.cond_free_k * (1 - i / N)))
            st.sqrt_recip = sched.f32(sched.sqrt_recip_ac, i)
            st.sqrt_recipm1 = sched.f32(sched.sqrt_recipm1_ac, i)
            st.coef1 = sched.f32(sched.coef1, i)
            st.coef2 = sched.f32(sched.coef2, i)
            st.nonzero = 0.0 if i == 0 else 1.0
        return steps, order

    def sample(self, sched: Schedule, x_T, step_noise):
        """x_T f32 [1, 100, S]; step_noise f32 [N, 1, 100, S] with step_noise[i] the draw of spaced index i
        (same convention as the oracle).  Returns the denormalised mel [1, 100, S]."""
        if self.S <= 0:
            raise ValueError("sample() needs condition() first (after sample_many the handle holds a batch)")
        N = sched.num_timesteps
        steps, order = self._run_order_steps(sched)
        x = x_T[0].to(self.device).float().contiguous()
        noise = step_noise.to(self.device).float()[order, 0].contiguous()  # run order
        mel = torch.empty(self.cfg.in_channels, self.S, device=self.device, dtype=torch.float32)
        E.check(self.lib.tt_diff_sample(self.h, E.ptr(x), E.ptr(noise), steps, N, int(sched.cond_free), E.ptr(mel), E.stream_ptr()))
        return mel[None]

    def sample_many(self, sched: Schedule, items):
        """Several utterances through one denoiser pass per step (tt_diff_sample_batch).  items: list of
        (latents f32 [1, M_u, latent], cond_latent f32 [1, 2C], S_u, x_T f32 [1, 100, S_u], step_noise f32 [N, 1, 100, S_u]); all of them
        walk `sched`.  Returns the list of denormalised mels [1, 100, S_u].  Each utterance is treated exactly as if it ran alone
        (its own statistics / attention span / zero padding); only the accumulation grouping of the GroupNorm partial sums
        differs from sample(), i.e. results agree within the operand tolerance, not bit for bit."""
        U = len(items)
        if not 1 <= U <= self.max_batch:
            raise ValueError(f"{U} utterances exceed this stage's batch capacity {self.max_batch}")
        N = sched.num_timesteps
        steps, order = self._run_order_steps(sched)
        S_pad = max(int(it[2]) for it in items)
        E.check(self.lib.tt_diff_batch_begin(self.h, U, S_pad, E.stream_ptr()))
        keep = []
        for u, (lat, cond, S, x_T, noise) in enumerate(items):
            lat_ = lat[0].to(self.device).float().contiguous()
            cond_ = cond[0].to(self.device).float().contiguous()
            idx = torch.from_numpy(nearest_interp_index(lat_.shape[0], S)).to(self.device)
            E.check(self.lib.tt_diff_condition_slot(self.h, u, E.ptr(lat_), lat_.shape[0], E.ptr(cond_), E.ptr(idx), int(S), E.stream_ptr()))
            x = x_T[0].to(self.device).float().contiguous()
            nz = noise.to(self.device).float()[order, 0].contiguous()
            mel = torch.empty(self.cfg.in_channels, int(S), device=self.device, dtype=torch.float32)
            keep.append((lat_, cond_, idx, x, nz, mel))
        arr = lambda k: (C.c_void_p * U)(*[E.ptr(t[k]) for t in keep])
        E.check(self.lib.tt_diff_sample_batch(self.h, U, arr(3), arr(4), steps, N, int(sched.cond_free), arr(5), E.stream_ptr()))
        self.S = 0  # the handle holds a batch: condition() again before sample()
        return [t[5][None] for t in keep]

    # ---- split sampling: this engine evaluates ONE denoiser row per step ---------------------
    def split_begin(self, sched: Schedule, x_T, step_noise, row):
        """row 0 = conditioned, 1 = conditioning-free.  Keeps the run-order noise and the output buffers alive."""
        if not sched.cond_free:
            raise ValueError("split sampling needs conditioning_free (two rows per step)")
        if self.S <= 0:
            raise ValueError("split sampling needs condition() first (the handle holds no single-utterance conditioning)")
        steps, order = self._run_order_steps(sched)
        x = x_T[0].to(self.device).float().contiguous()
        self._split_noise = step_noise.to(self.device).float()[order, 0].contiguous()
        self._split_mel = torch.empty(self.cfg.in_channels, self.S, device=self.device, dtype=torch.float32)
        self._split_row = torch.empty(self.S, self.cfg.out_channels, device=self.device, dtype=torch.float32)
        E.check(self.lib.tt_diff_split_begin(self.h, E.ptr(x), steps, sched.num_timesteps, int(row), E.stream_ptr()))
        torch.cuda.current_stream().synchronize()  # `x` and `steps` are consumed
        return sched.num_timesteps

    def split_forward(self):
        """This engine's model row of the current step: f32 [S, out_channels] (valid until the next call)."""
        E.check(self.lib.tt_diff_split_forward(self.h, E.ptr(self._split_row), E.stream_ptr()))
        return self._split_row

    def split_update(self, rows):
        """rows f32 [2, S, out_channels]: row 0 conditioned, row 1 conditioning-free (from both participants)."""
        assert rows.shape == (2, self.S, self.cfg.out_channels) and rows.is_contiguous() and rows.dtype == torch.float32
        E.check(self.lib.tt_diff_split_update(self.h, E.ptr(rows), E.ptr(self._split_noise), E.ptr(self._split_mel), E.stream_ptr()))

    def split_end(self):
        E.check(self.lib.tt_diff_split_end(self.h))
        mel = self._split_mel[None]
        self._split_noise = self._split_row = self._split_mel = None
        return mel

    def sample_split(self, sched: Schedule, x_T, step_noise, row, exchange):
        """p_sample_loop with the two rows on two participants.  `exchange(rows, mine)` fills rows[r] with
        participant r's `mine` (an all_gather over the pair, tortoise_tts_amd/dist.py)."""
        n = self.split_begin(sched, x_T, step_noise, row)
        rows = torch.empty(2, self.S, self.cfg.out_channels, device=self.device, dtype=torch.float32)
        for _ in range(n):
            exchange(rows, self.split_forward())
            self.split_update(rows)
        return self.split_end()


class ConditioningStage:
    """Conditioning encoders of the voice_samples path on the device (csrc/cond.hip):
    UnifiedVoice.get_conditioning (ConditioningEncoder, autoregressive.py:204-228, 444-452) and
    DiffusionTts.get_conditioning (contextual_embedder, diffusion_decoder.py:186-192, 222-230).  Inputs are the mel
    spectrograms api.py:271-289 builds from the clips; the per-clip results are combined exactly as the reference does."""

    def __init__(self, sd_ar, sd_diff, ar_cfg, diff_cfg, device="cuda", dtype=E.TT_BF16, max_frames=1024):
        self.lib = E.init()
        self.device = torch.device(device)
        self.ar_cfg, self.diff_cfg = ar_cfg, diff_cfg
        self.w = pack.pack_conditioning(sd_ar, sd_diff, ar_cfg, diff_cfg, self.device, dtype)
        sh = self.w.shape
        c = E.CondConfig()
        c.dtype = dtype
        c.ar_dim, c.ar_heads, c.ar_blocks = ar_cfg.model_dim, ar_cfg.heads, sh["ar_blocks"]
        c.ar_mel, c.ar_mel_pad = sh["ar_mel"], sh["mel_pad"]
        c.diff_channels, c.diff_heads, c.diff_blocks = sh["diff_channels"], sh["diff_heads"], sh["diff_blocks"]
        c.diff_mel, c.diff_mel_pad = sh["diff_mel"], sh["mel_pad"]
        c.max_frames = max_frames
        self.cfg = c
        self.handle = C.c_void_p()
        E.check(self.lib.tt_cond_create(C.byref(c), C.byref(self.w.weights), C.byref(self.handle)))

    def close(self):
        if self.handle:
            self.lib.tt_cond_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _clips(self, mels, n_mel):
        """Accepts [1, n_clips, n_mel, T] (the reference's stacked tensor) or a list of [1, n_mel, T] / [n_mel, T] clips."""
        if torch.is_tensor(mels):
            mels = [mels[0, j] for j in range(mels.shape[1])] if mels.dim() == 4 else [mels]
        out = []
        for m in mels:
            m = m.to(self.device).float()
            m = m.reshape(-1, m.shape[-1]) if m.dim() == 3 else m
            if m.shape[0] != n_mel:
                raise ValueError(f"conditioning clip has {m.shape[0]} mel channels, the encoder takes {n_mel}")
            out.append(m.contiguous())
        if not out:
            raise ValueError("no conditioning clips")
        return out

    def auto_latent(self, mels):
        """mels f32 [1, n_clips, 80, T] (or a list of clips) -> f32 [1, model_dim]: mean over the clips of h[:, :, 0]."""
        clips = self._clips(mels, self.cfg.ar_mel)
        acc = torch.zeros(self.cfg.ar_dim, device=self.device, dtype=torch.float32)
        out = torch.empty(self.cfg.ar_dim, device=self.device, dtype=torch.float32)
        for m in clips:
            E.check(self.lib.tt_cond_ar_clip(self.handle, E.ptr(m), m.shape[1], E.ptr(out), E.stream_ptr()))
            acc += out
        return (acc / len(clips))[None]

    def diffusion_latent(self, mels):
        """mels f32 [1, n_clips, 100, T] (or a list of clips) -> f32 [1, 2 * model_channels]: the clips' embedder outputs
        concatenated along time and averaged."""
        clips = self._clips(mels, self.cfg.diff_mel)
        C2 = 2 * self.cfg.diff_channels
        acc = torch.zeros(C2, device=self.device, dtype=torch.float32)
        out = torch.empty(C2, device=self.device, dtype=torch.float32)
        total = 0
        for m in clips:
            frames = C.c_int(0)
            E.check(self.lib.tt_cond_diff_clip(self.handle, E.ptr(m), m.shape[1], E.ptr(out), C.byref(frames), E.stream_ptr()))
            acc += out
            total += frames.value
        return (acc / total)[None]


class RandomLatentStage:
    """get_random_conditioning_latents (api.py:301-309): the two RandomLatentConverter MLPs as six M = 1 GEMMs each.
    EqualLinear's constants are folded at pack time: leaky_relu is positively homogeneous, so
    leaky_relu(x W^T s + b l) * sqrt2 == leaky_relu(x (W s sqrt2)^T + b l sqrt2)."""

    def __init__(self, sd_auto, sd_diffuser, device="cuda", dtype=E.TT_BF16, lr_mul=0.1):
        self.lib = E.init()
        self.device = torch.device(device)
        self.dtype = dtype
        self.h = pack.Holder(self.device, dtype)
        self.nets = []
        for sd in (sd_auto, sd_diffuser):
            C_ = sd["layers.0.weight"].shape[0]
            scale = (1.0 / np.sqrt(C_)) * lr_mul * np.sqrt(2.0)
            layers = [(self.h.op(sd[f"layers.{i}.weight"].float() * scale), self.h.f32(sd[f"layers.{i}.bias"].float() * (lr_mul * np.sqrt(2.0))))
                      for i in range(5)]
            layers.append((self.h.op(sd["layers.5.weight"]), self.h.f32(sd["layers.5.bias"])))
            self.nets.append((C_, layers))
        self.channels = tuple(n[0] for n in self.nets)

    def _run(self, net, r):
        C_, layers = net
        x = r.to(self.device).float().reshape(1, C_).to(self.h.tdtype).contiguous()
        out = None
        for i, (w, b) in enumerate(layers):
            last = i == len(layers) - 1
            nxt = None if last else torch.empty(1, C_, device=self.device, dtype=self.h.tdtype)
            out = torch.empty(1, C_, device=self.device, dtype=torch.float32) if last else None
            E.check(self.lib.tt_op_gemm(self.dtype, E.ptr(x), C_, E.ptr(w), C_, 1, C_, C_, 1, 0, 1, E.ptr(b),
                                        E.ACT_NONE if last else E.ACT_LRELU, None, E.ptr(out), E.ptr(nxt), E.stream_ptr()))
            x = nxt
        return out

    def latents(self, r_auto, r_diffuser):
        """r_auto f32 [1, 1024], r_diffuser f32 [1, 2048] standard-normal draws -> (auto latent, diffusion latent)."""
        return self._run(self.nets[0], r_auto), self._run(self.nets[1], r_diffuser)


class HifiganStage:
    """HiFi-GAN decoder of the streaming path (csrc/hifigan.hip): hifi_decoder.inference(gpt_latents,
    auto_conditioning) of api_fast.py:420 / 517 (hifigan_decoder.py:259-289)."""

    def __init__(self, sd_folded, cfg, device="cuda", dtype=E.TT_BF16, max_latents=512):
        self.lib = E.init()
        self.device = torch.device(device)
        self.cfg = cfg
        self.w = pack.pack_hifigan(sd_folded, cfg, self.device, dtype)
        c = E.HifiConfig()
        c.dtype = dtype
        c.in_channels, c.cond_channels, c.initial_channel = cfg.in_channels, cfg.cond_channels, cfg.upsample_initial_channel
        c.num_stages = len(cfg.upsample_factors)
        for i, u in enumerate(cfg.upsample_factors):
            c.up_factor[i] = u
        c.num_kernels = len(cfg.resblock_kernel_sizes)
        for i, k in enumerate(cfg.resblock_kernel_sizes):
            c.kernel_size[i] = k
        c.num_dilations = len(cfg.resblock_dilation_sizes)
        for i, d in enumerate(cfg.resblock_dilation_sizes):
            c.dilation[i] = d
        c.lrelu_slope = cfg.lrelu_slope
        c.max_latents = max_latents
        self.c = c
        self.handle = C.c_void_p()
        E.check(self.lib.tt_hifi_create(C.byref(c), C.byref(self.w.weights), C.byref(self.handle)))

    def close(self):
        if self.handle:
            self.lib.tt_hifi_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def inference(self, latents, g):
        """latents f32 [1, T, in_channels], g f32 [1, cond_channels] -> wav f32 [1, 1, frames * hop] (on the device)."""
        lat = latents.to(self.device).float().reshape(-1, self.cfg.in_channels).contiguous()
        gv = g.to(self.device).float().reshape(-1).contiguous()
        T = lat.shape[0]
        n = self.lib.tt_hifi_output_frames(T) * self.cfg.hop
        wav = torch.empty(n, device=self.device, dtype=torch.float32)
        ns = C.c_int(0)
        E.check(self.lib.tt_hifi_run(self.handle, E.ptr(lat), T, E.ptr(gv), E.ptr(wav), C.byref(ns), E.stream_ptr()))
        assert ns.value == n, (ns.value, n)
        return wav[None, None]


class VocoderStage:
    """UnivNetGenerator.inference (vocoder.py:300-312)."""

    def __init__(self, sd_folded, cfg: VocoderConfig = VocoderConfig(), device="cuda", dtype=E.TT_BF16, max_frames=2304):
        self.lib = E.init()
        self.cfg = cfg
        self.device = torch.device(device)
        self.w = pack.pack_vocoder(sd_folded, cfg, self.device, dtype)
        c = E.VocConfig()
        c.dtype, c.max_frames, c.mel_channels, c.mel_pad = dtype, max_frames, cfg.n_mel_channels, self.w.mel_pad
        self.h = E.vp()
        E.check(self.lib.tt_voc_create(C.byref(c), C.byref(self.w.weights), C.byref(self.h)))

    def close(self):
        if self.h:
            self.lib.tt_voc_destroy(self.h)
            self.h = E.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def guard(self, reset=True):
        """Workgroups that met non-finite predicted LVC kernels since the last reset (operand-overflow guard; read after a synchronisation)."""
        return E.guard_count(self.lib.tt_voc_guard(self.h, int(reset)))

    def inference(self, mel, z):
        """mel f32 [1, 100, S]; z f32 [1, 64, S+10] -> audio [1, 1, S*256]."""
        m = mel[0].to(self.device).float().contiguous()
        S = m.shape[1]
        zz = z[0].to(self.device).float().contiguous()
        assert zz.shape == (self.cfg.noise_dim, S + 10)
        audio = torch.empty(S * self.cfg.hop_length, device=self.device, dtype=torch.float32)
        E.check(self.lib.tt_voc_run(self.h, E.ptr(m), S, E.ptr(zz), E.ptr(audio), E.stream_ptr()))
        return audio.clamp(-1, 1)[None, None]
