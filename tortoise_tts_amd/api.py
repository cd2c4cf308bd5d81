"""Drop-in surface: `TextToSpeech` with the reference's constructor / tts() / tts_with_preset()
signatures (reference: tortoise/api.py:179-181, 311-332, 334-342), re-hosted on the MI355X engine.

What changed behind the signature
  * all four networks stay resident on the GPU (the reference re-uploads ~5.6 GB per call through
    `temporary_cuda`, api.py:245-249);
  * stage 1 decodes all candidates of this rank together, sharing one prefix evaluation, with
    on-device sampling and a hipGraph per token;
  * candidates shard across the GPUs of a node, one all_gather picks the CLVP top-k (dist.py);
  * stage 2 evaluates conditioned + unconditioned denoiser rows in one pass per step;
  * integer post-processing (api.py:87-114, 547-556) is vectorised on device and bit-exact.
Constructor flags of the reference and what they mean here:
  * kv_cache   the engine always keeps a KV cache; the flag selects the reference's mel POSITION rule, which is the
               only numerical difference between its two code paths (kv_cache=False, the reference default: rows
               0,1,2,...; kv_cache=True: rows 0,2,3,...; autoregressive.py:134-149);
  * half       True selects fp16 MFMA operands (the reference's fp16 autocast), False the engine default (bf16) unless
               the engine-only `dtype=` says otherwise;
  * enable_redaction  bracketed text needs the wav2vec2 aligner (out of scope): such text raises, other text is unaffected.
cvvp_amount > 0 (api.py:450-472; the CHANGELOG calls CVVP "removed", the call sites and cvvp.pth remain): the CVVP model is built on
first use like upstream (load_cvvp -> stages.CvvpStage, csrc/cvvp.hip) and blended into the CLVP ranking when voice_samples are given.
Out of scope (raise, never silently fall back): wav2vec redaction of bracketed text, DeepSpeed flag.
"""
import os
import random

import numpy as np
import torch
import torch.nn.functional as F

from . import dist as tdist
from . import engine as E
from . import stages
from . import weights as W
from .config import ARConfig, CLVPConfig, CVVPConfig, DiffusionConfig, VocoderConfig, PRESETS, BASE_SETTINGS, CALM_TOKEN
from .schedule import Schedule

MODELS_DIR = os.environ.get("TORTOISE_MODELS_DIR", os.path.join(os.path.expanduser("~"), ".cache", "tortoise", "models"))
MODEL_FILES = {"autoregressive": "autoregressive.pth", "clvp": "clvp2.pth", "diffusion": "diffusion_decoder.pth",
               "vocoder": "vocoder.pth", "rlg_auto": "rlg_auto.pth", "rlg_diffuser": "rlg_diffuser.pth", "cvvp": "cvvp.pth"}


class _StageTimer:
    """Stage boundaries of one tts() call as HIP events on the current stream (no host synchronisation until read)."""

    def __init__(self, n):
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(n)]

    def mark(self, i):
        self.ev[i].record()

    def seconds(self, i, j):
        return self.ev[i].elapsed_time(self.ev[j]) / 1e3

    @staticmethod
    def synchronize():
        torch.cuda.synchronize()


def sampler_kwargs(hf_generate_kwargs):
    """The **hf_generate_kwargs of tts() the on-device sampler honours -> (top_k, typical_mass); anything else raises instead of
    being dropped.  `top_k`: HF GenerationConfig default 50, which the reference inherits (api.py never overrides it).
    `typical_sampling` / `typical_mass` (api.py:361-364): inference_speech turns them into generate()'s logits_processor list
    [TypicalLogitsWarper(mass=typical_mass)] (autoregressive.py:536, 558); typical_mass is ignored unless typical_sampling is set,
    as upstream.  Returns typical_mass = 0.0 for "off"."""
    kw = dict(hf_generate_kwargs)
    top_k = int(kw.pop("top_k", 50))
    typical = bool(kw.pop("typical_sampling", False))
    mass = float(kw.pop("typical_mass", .9))
    if kw:
        raise NotImplementedError(f"unsupported generate kwargs {sorted(kw)}: the on-device sampler implements temperature / top_k / top_p / "
                                  f"repetition_penalty / typical_sampling + typical_mass (length_penalty is a no-op when sampling)")
    if typical and not 0.0 < mass < 1.0:
        raise ValueError(f"typical_mass={mass} must lie in (0, 1) (the reference's warper indexes past the vocabulary at >= 1)")
    return top_k, (mass if typical else 0.0)


def fix_autoregressive_output(codes, stop_token, calm_token=CALM_TOKEN):
    """Vectorised api.py:87-114 for a batch int tensor [B, n] (any device); rows without a stop token are
    returned unchanged exactly like the reference (which only prints a warning)."""
    codes = codes.clone()
    B, n = codes.shape
    is_stop = codes == stop_token
    has_stop = is_stop.any(dim=1)
    pos = torch.arange(n, device=codes.device)[None, :].expand(B, n)
    first = torch.where(is_stop, pos, torch.full_like(pos, n)).min(dim=1).values  # stm
    tail = (pos >= first[:, None]) & has_stop[:, None]
    codes[tail] = calm_token
    rows = has_stop & (first - 3 < n)
    codes[rows, -3] = 45
    codes[rows, -2] = 45
    codes[rows, -1] = 248
    return codes


def calm_trim_length(codes_row, calm_token=CALM_TOKEN):
    """api.py:547-556: index k at which more than 8 consecutive calm tokens have been seen (latents are cut
    to [:k]); len(codes_row) if there is no such run.  Vectorised, one host read."""
    c = (codes_row == calm_token).to(torch.int32)
    n = c.shape[0]
    if n < 9:
        return n
    run9 = F.avg_pool1d(c[None, None].float(), kernel_size=9, stride=1)[0, 0] >= 1.0 - 1e-6  # windows of 9 calm tokens
    idx = torch.nonzero(run9)
    return int(idx[0, 0]) + 8 if idx.numel() else n


def _load_state_dict(models_dir, name):
    path = os.path.join(models_dir, MODEL_FILES[name])
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found. Put the reference checkpoints in models_dir (or $TORTOISE_MODELS_DIR), or pass "
                                f"state_dicts= to TextToSpeech; there is no network access to download them.")
    sd = torch.load(path, map_location="cpu")
    return sd["model_g"] if name == "vocoder" else sd


def _load_file(models_dir, filename):
    path = os.path.join(models_dir, filename)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found. Put the reference checkpoints in models_dir (or $TORTOISE_MODELS_DIR), or pass "
                                f"state_dicts= to TextToSpeech; there is no network access to download them.")
    return torch.load(path, map_location="cpu")


def utterance_batch_bytes(utterance_batch, max_candidates, max_mel_tokens, ar_cfg, diff_cfg, max_steps=512):
    """Bytes of the two arenas that grow with TextToSpeech(utterance_batch=): the per-sequence KV cache (layers x model_dim x K,V x 2
    bytes per cached token, utterance_batch x max_candidates sequences of max_mel_tokens + 2 slots) and the conditioning-integrator outputs
    of every sampler step (max_steps x 2 guidance rows x utterances x S x channels x 2 bytes)."""
    max_S = max_mel_tokens * 4 * 24000 // 22050 + 8
    kv = utterance_batch * max_candidates * (max_mel_tokens + 2) * ar_cfg.layers * ar_cfg.model_dim * 2 * 2
    integ = max_steps * 2 * utterance_batch * max_S * diff_cfg.model_channels * 2
    return kv + integ


STAGE_NAMES = ("ar", "clvp", "diffusion", "vocoder")


def resolve_stage_dtypes(dtype, half):
    """MFMA operand type per stage -> {'ar' | 'clvp' | 'diffusion' | 'vocoder': engine dtype code}.

    The reference autocasts ONLY the autoregressive + CLVP stages to fp16, and only under half=True (api.py:413-414, 460-463); its
    diffusion decoder and vocoder always run in fp32 (api.py:225 use_fp16=False, 540-560 no autocast).  Default here (round 6): **fp16 for
    every stage** - the MFMA operand type closest to the reference's fp32 (3 more mantissa bits than bf16) at the same speed.  Measured
    against the reference's own fp32 modules at the benchmarked width (profiles/r06_parity_gpu.txt): decode logits over
    500 teacher-forced steps rel-L2 8e-4 (bf16 6.5e-3), total variation of the sampler's warped distribution mean 0.002 / max 0.020
    (bf16 0.009 / 0.038), CLVP Spearman 0.9999 (0.9993), 200-iteration mel 1.1e-3 (8.6e-3).  fp16 saturates at 65504, so every stage
    counts non-finite values behind its operand casts (tt_*_guard): a tripped stage is rebuilt with bf16 operands - the fp32 exponent
    range - and the utterance re-rendered with the same seed (TextToSpeech._demote).  half=True (the reference's fp16 autocast flag) is
    therefore the default behaviour already; dtype='bf16' (or a dict per stage) selects bf16 operands up front.
    `dtype` may be None (defaults), one name for every stage, or a dict overriding some stages."""
    out = {"ar": "fp16", "clvp": "fp16", "diffusion": "fp16", "vocoder": "fp16"}
    if isinstance(dtype, dict):
        unknown = set(dtype) - set(STAGE_NAMES)
        if unknown:
            raise ValueError(f"dtype: unknown stage(s) {sorted(unknown)}; stages are {STAGE_NAMES}")
        out.update(dtype)
    elif dtype is not None:
        out = {k: dtype for k in STAGE_NAMES}
    if half and any(E.dtype_code(out[k]) != E.TT_F16 for k in ("ar", "clvp")):
        raise ValueError(f"half=True asks for fp16 operands in the autoregressive / CLVP stages but dtype={dtype!r} was also given")
    return {k: E.dtype_code(v) for k, v in out.items()}


class TextToSpeech:
    """Main entry point; see the module docstring.  Engine-only keyword arguments (all optional, after
    the reference's): `state_dicts` (dict of reference-layout state_dicts instead of files in
    models_dir), `dtype` (MFMA operand type: 'bf16' | 'fp16' for every stage, or a dict per stage
    {'ar', 'clvp', 'diffusion', 'vocoder'}; see resolve_stage_dtypes for the defaults), `max_candidates`
    (per-GPU decode batch capacity), `configs` (ARConfig/CLVPConfig/DiffusionConfig/VocoderConfig overrides for tests)."""

    def __init__(self, autoregressive_batch_size=None, models_dir=MODELS_DIR, enable_redaction=True, kv_cache=False,
                 use_deepspeed=False, half=False, device=None, tokenizer_vocab_file=None, tokenizer_basic=False, *,
                 state_dicts=None, dtype=None, max_candidates=256, configs=None, max_mel_tokens=500, max_text_tokens=402,
                 candidate_sharding=True, utterance_batch=1):
        self.models_dir = models_dir
        if use_deepspeed:
            raise NotImplementedError("use_deepspeed: DeepSpeed kernel injection is a CUDA-only reference option; the MI355X engine "
                                      "always runs its own fused HIP path")
        # wav2vec2 redaction is outside the hot path.  The flag is kept (reference default True): text
        # without [brackets] is unaffected by it in the reference too; bracketed text raises in tts() instead of being spoken.
        self.enable_redaction = bool(enable_redaction)
        self.kv_cache = bool(kv_cache)
        self.half = bool(half)
        # candidate_sharding=False: this instance renders whole utterances on its own GPU even inside a multi-rank job (the
        # long-form driver spreads CHUNKS over the ranks instead: tortoise_tts_amd/longform.py, BASELINE config #4)
        self.rank, self.world = tdist.world() if candidate_sharding else (0, 1)
        # With >= 2 ranks a single winner's diffusion tail is split over ranks 0 and 1 (conditioned / conditioning-free
        # row each, one exchange per step).  TT_SPLIT_DIFFUSION=0 keeps the whole tail on rank 0.
        self.split_diffusion = self.world >= 2 and os.environ.get("TT_SPLIT_DIFFUSION", "1") != "0"
        if self.split_diffusion:
            tdist.pair_group()  # collective over all ranks: create it once, here, where every rank passes
        self.device = E.require_gpu(device)
        self.dtypes = resolve_stage_dtypes(dtype, half)  # half=True is the reference's fp16 autocast (api.py:180, autoregressive.py:561)
        self.dtype = self.dtypes["ar"]  # (conditioning encoders / random-latent MLPs follow the autoregressive stage)
        self.demotions = []             # stages whose overflow guard tripped and that were rebuilt with bf16 operands
        cfgs = configs or {}
        self.ar_cfg = cfgs.get("ar", ARConfig())
        self.clvp_cfg = cfgs.get("clvp", CLVPConfig())
        self.diff_cfg = cfgs.get("diffusion", DiffusionConfig())
        self.voc_cfg = cfgs.get("vocoder", VocoderConfig())
        self.cvvp_cfg = cfgs.get("cvvp", CVVPConfig())
        sds = state_dicts or {}

        def sd(name):
            return sds[name] if name in sds else _load_state_dict(models_dir, name)

        # the reference's default AR batch is 16 on a >=14 GB GPU (api.py:148-172); 288 GB of HBM3E decodes
        # every candidate of this rank in one batch unless the caller asks for smaller batches.
        cap = min(int(autoregressive_batch_size or max_candidates), max_candidates)
        self.autoregressive_batch_size = cap  # never above the handle's decode capacity
        self.tokenizer_args = (tokenizer_vocab_file, tokenizer_basic)
        self._tokenizer = None
        self.max_mel_tokens_cap = max_mel_tokens
        max_S = max_mel_tokens * 4 * 24000 // 22050 + 8
        # utterance_batch > 1 (tts_many, long-form reading): that many utterances share ONE decode batch - the KV caches of
        # utterance_batch x max_candidates sequences are resident (122 880 B per cached token: 8 x 256 x 200 tokens = 50 GB of the 288)
        self.utterance_batch = max(1, int(utterance_batch))
        if self.utterance_batch > 16:
            raise ValueError("utterance_batch is limited to 16 utterances per decode batch")
        if self.utterance_batch > 1 and torch.device(self.device).type == "cuda":
            need = utterance_batch_bytes(self.utterance_batch, cap, max_mel_tokens, self.ar_cfg, self.diff_cfg)
            total = torch.cuda.get_device_properties(self.device).total_memory
            if need > 0.8 * total:
                raise ValueError(f"utterance_batch={self.utterance_batch} x max_candidates={cap} x max_mel_tokens={max_mel_tokens} needs "
                                 f"{need / 2 ** 30:.0f} GiB of KV cache + integrator slices, the device has {total / 2 ** 30:.0f} GiB: "
                                 f"lower utterance_batch or max_mel_tokens")
        self._caps = dict(cap=cap, max_text_tokens=max_text_tokens, max_mel_tokens=max_mel_tokens, max_S=max_S)
        self._state_dicts = sds
        # tts_many also pushes utterance_batch utterances through ONE denoiser pass per diffusion step (padded to the longest)
        self.batch_diffusion = self.utterance_batch > 1
        # (Decoding batch i + 1 on a second stream while batch i's denoiser passes run was built and measured: both phases slow down by
        # what the other takes - decode 3.1 -> 4.35 s, rendering 2.85 -> 3.93 s for 15 chunks at 8 per batch, 6.4 -> 6.19 s in total, and
        # one batch of 16 is faster still, 6.06 s: profiles/r03_bench_read_overlap.txt - and removed again.)
        for name in STAGE_NAMES:
            self._build_stage(name)
        self.rlg = None         # random-voice latent MLPs, built lazily like the reference (api.py:301-309)
        self.cvvp = None        # "CVVP model is only loaded if used" (api.py:234): built by the first tts(cvvp_amount > 0)
        self.conditioning = None  # conditioning encoders (voice_samples path), built lazily
        self.mel_front_end = None
        # attributes the reference exposes and callers touch (api.py:408, 523)
        self.stop_mel_token = self.ar_cfg.stop_mel_token
        self.mel_length_compression = self.ar_cfg.mel_length_compression
        self.timings = {}

    def set_candidate_sharding(self, on):
        """Switch an instance inside a multi-rank job between sharding the candidates of ONE utterance over the ranks (on: one
        all_gather per utterance, the winner rendered by ranks 0 / 1) and rendering whole utterances on its own GPU (off: replicas, no
        data-path collective).  The engines need max_candidates >= the candidates a call decodes on this rank in either mode."""
        self.rank, self.world = tdist.world() if on else (0, 1)
        self.split_diffusion = self.world >= 2 and os.environ.get("TT_SPLIT_DIFFUSION", "1") != "0"
        if self.split_diffusion:
            tdist.pair_group()

    def _build_stage(self, name):
        """(Re)build one stage engine with self.dtypes[name] operands."""
        c, dt = self._caps, self.dtypes[name]
        old = getattr(self, {"ar": "ar", "clvp": "clvp", "diffusion": "diffusion", "vocoder": "vocoder"}[name], None)
        if old is not None:
            old.close()
        if name == "ar":
            # (capacity of at least 8 sequences: handles of <= 4 are streaming-size handles whose decode GEMMs run GEMV-shaped - other bits than
            #  the MFMA path - and a candidate's codes must not depend on how few candidates a rank happens to get: csrc/gemv.hip)
            self.ar = stages.ArStage(self._sd("autoregressive"), self.ar_cfg, self.device, dt, max_batch=max(c["cap"], 8) * self.utterance_batch,
                                     max_text=c["max_text_tokens"], max_new_tokens=c["max_mel_tokens"], max_latent_candidates=4,
                                     kv_cache=self.kv_cache, max_groups=self.utterance_batch)
        elif name == "clvp":
            # (tts_many ranks the utterances of a wave in ONE speech-tower pass: capacity for utterance_batch x cap candidates)
            self.clvp = stages.ClvpStage(self._sd("clvp"), self.clvp_cfg, self.device, dt,
                                         max_rows=max(c["cap"], 8) * c["max_mel_tokens"] * self.utterance_batch)
            if getattr(self, "cvvp", None) is not None:  # CVVP runs under the same autocast as CLVP in the reference (api.py:447-449): same operand type
                self.cvvp.close()
                self.cvvp = None
                self.load_cvvp()
        elif name == "diffusion":
            self.diffusion = stages.DiffusionStage(self._sd("diffusion"), self.diff_cfg, self.device, dt, max_seq=c["max_S"],
                                                   max_codes=c["max_mel_tokens"] + 8, max_steps=512, max_batch=self.utterance_batch)
        elif name == "vocoder":
            voc_sd = self._sd("vocoder")
            if any(k.endswith("weight_v") for k in voc_sd):
                voc_sd = W.fold_weight_norm(voc_sd)  # UnivNetGenerator.eval(inference=True), vocoder.py:284-298
            self.vocoder = stages.VocoderStage(voc_sd, self.voc_cfg, self.device, dt, max_frames=c["max_S"])
        else:
            raise ValueError(name)

    def load_cvvp(self):
        """api.py:252-256: the CVVP model (cvvp.pth) as a device stage, with the CLVP stage's operand type."""
        if self.cvvp is None:
            c = self._caps
            self.cvvp = stages.CvvpStage(self._sd("cvvp"), self.cvvp_cfg, self.device, self.dtypes["clvp"],
                                         max_rows=max(c["cap"], 8) * c["max_mel_tokens"])
        return self.cvvp

    def _tripped_stages(self, wav_ok=True):
        """Stages whose operand-overflow guard counted non-finite values during the utterance that just finished (the caller has
        synchronised); agreed over the ranks so that every rank takes the same decision.  Resets the counters."""
        flags = []
        for name in ("ar", "clvp", "diffusion"):
            g = getattr(getattr(self, name), "guard", None)
            flags.append(bool(g()) if g is not None else False)
            if name == "clvp" and self.cvvp is not None and getattr(self.cvvp, "guard", None) is not None:
                flags[-1] = bool(self.cvvp.guard()) or flags[-1]  # (always read: reading resets the counter)
        vg = getattr(self.vocoder, "guard", None)  # non-finite predicted LVC kernels: the gate and the final tanh would hide them from wav_ok
        voc_tripped = bool(vg()) if vg is not None else False  # (always read: reading resets the counter)
        flags.append((not wav_ok) or voc_tripped)
        if self.world > 1:
            flags = tdist.any_over_ranks(flags)
        # only the FIRST tripped stage in pipeline order is at fault for certain: the later ones may merely have been fed its
        # non-finite output (an overflowed denoiser hands the vocoder a NaN mel); they get their own turn after the re-render
        return [n for n, f in zip(STAGE_NAMES, flags) if f][:1]

    def _demote(self, tripped):
        """fp16 stages among `tripped` are rebuilt with bf16 operands (same weights, fp32 exponent range); a stage that overflows
        in bf16 has non-finite weights or inputs: that is an error, not a precision choice."""
        import warnings
        for name in tripped:
            if self.dtypes[name] != E.TT_F16:
                raise E.OperandOverflow(f"the {name} stage produced non-finite values with bf16 operands (non-finite weights or inputs?)")
            warnings.warn(f"tortoise_tts_amd: the {name} stage overflowed fp16 operands; rebuilding it with bf16 operands and re-rendering")
            self.dtypes[name] = E.TT_BF16
            self.demotions.append(name)
            self._build_stage(name)

    # ------------------------------------------------------------------ reference helpers
    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from .text import VoiceBpeTokenizer
            self._tokenizer = VoiceBpeTokenizer(self.tokenizer_args[0], self.tokenizer_args[1], self.models_dir)
        return self._tokenizer

    def get_conditioning_latents(self, voice_samples, return_mels=False):
        """api.py:258-299 on the engine: ConditioningEncoder (autoregressive.py:204-228) and contextual_embedder
        (diffusion_decoder.py:186-192, 222-230) run on the device (csrc/cond.hip).  voice_samples is, as in
        the reference, a list of 22.05 kHz waveform tensors (the mel front-end of api.py:271-287 then runs in torch,
        tortoise_tts_amd/audio.py: restated without torchaudio / librosa, pinned through oracle/audio_oracle.py), or a list of ready
        (auto_mel f32 [1, 80, T_a], diffusion_mel f32 [1, 100, T_d]) pairs, one per clip."""
        if torch.is_tensor(voice_samples):
            voice_samples = [voice_samples]  # api.py:269-270
        if self.conditioning is None:
            self.conditioning = stages.ConditioningStage(self._sd("autoregressive"), self._sd("diffusion"), self.ar_cfg, self.diff_cfg,
                                                         self.device, self.dtype)
        auto_mels, diff_mels = [], []
        for vs in voice_samples:
            if isinstance(vs, (tuple, list)) and len(vs) == 2:
                am, dm = vs
            elif torch.is_tensor(vs):
                if self.mel_front_end is None:
                    from .audio import MelFrontEnd
                    self.mel_front_end = MelFrontEnd(self.models_dir)
                am, dm = self.mel_front_end(vs.to(self.device))
            else:
                raise TypeError("voice_samples entries must be waveform tensors or (auto_mel, diffusion_mel) pairs")
            auto_mels.append(am.to(self.device).float().reshape(1, am.shape[-2], am.shape[-1]))
            diff_mels.append(dm.to(self.device).float().reshape(1, dm.shape[-2], dm.shape[-1]))
        auto_latent = self.conditioning.auto_latent(auto_mels)
        diffusion_latent = self.conditioning.diffusion_latent(diff_mels)
        if return_mels:
            return auto_latent, diffusion_latent, torch.stack(auto_mels, dim=1), torch.stack(diff_mels, dim=1)
        return auto_latent, diffusion_latent

    def _sd(self, name):
        return self._state_dicts[name] if name in self._state_dicts else _load_state_dict(self.models_dir, name)

    def get_random_conditioning_latents(self):
        """api.py:301-309: two RandomLatentConverter MLPs (random_latent_generator.py:42-55) on the device.  Like the
        reference, the Gaussian inputs are drawn from torch's CPU generator (it evaluates the MLPs on a CPU tensor), so the
        same torch.manual_seed gives the same random voice."""
        if self.rlg is None:
            self.rlg = stages.RandomLatentStage(self._sd("rlg_auto"), self._sd("rlg_diffuser"), self.device, self.dtype)
        ca, cd = self.rlg.channels  # 1024 / 2048 for the released rlg_auto.pth / rlg_diffuser.pth
        return self.rlg.latents(torch.randn(1, ca), torch.randn(1, cd))

    def dtype_names(self):
        """{'ar': 'bf16', ...}: the operand type every stage currently runs with (after any overflow demotion)."""
        return {k: E.DTYPE_NAMES[v] for k, v in self.dtypes.items()}

    def deterministic_state(self, seed=None):
        """api.py:598-609."""
        seed = int(torch.seed() % (2 ** 31)) if seed is None else int(seed)
        if self.world > 1:
            seed = tdist.broadcast_int(seed)  # every rank must draw the same noise and key the same Philox streams
        torch.manual_seed(seed)
        random.seed(seed)
        return seed

    def tts_with_preset(self, text, preset="fast", **kwargs):
        """api.py:311-332: same preset table, caller kwargs win."""
        settings = dict(BASE_SETTINGS)
        settings.update(PRESETS[preset])
        settings.update(kwargs)
        return self.tts(text, **settings)

    # ------------------------------------------------------------------ the pipeline
    @torch.no_grad()
    def tts(self, text, voice_samples=None, conditioning_latents=None, k=1, verbose=True, use_deterministic_seed=None,
            return_deterministic_state=False,
            # autoregressive generation parameters follow
            num_autoregressive_samples=512, temperature=.8, length_penalty=1, repetition_penalty=2.0, top_p=.8, max_mel_tokens=500,
            # CVVP parameters follow
            cvvp_amount=.0,
            # diffusion generation parameters follow
            diffusion_iterations=100, cond_free=True, cond_free_k=2, diffusion_temperature=1.0,
            **hf_generate_kwargs):
        noise = hf_generate_kwargs.pop("noise_override", None) or {}
        top_k, typical_mass = sampler_kwargs(hf_generate_kwargs)
        if not 0 <= cvvp_amount <= 1:
            raise ValueError(f"cvvp_amount={cvvp_amount} must lie in [0, 1] (api.py:366-367)")
        dev = self.device
        if self.enable_redaction and isinstance(text, str) and "[" in text and "]" in text:
            raise NotImplementedError("text with [bracketed] passages needs the wav2vec2 aligner to redact them from the audio "
                                      "(api.py:583-587), which is outside the accelerated path; remove the brackets or construct "
                                      "TextToSpeech(enable_redaction=False) to have them spoken")
        seed = self.deterministic_state(seed=use_deterministic_seed)
        ev = _StageTimer(6)
        ev.mark(0)

        if isinstance(text, str):
            text_tokens = torch.IntTensor(self.tokenizer.encode(text)).unsqueeze(0)
        else:  # pre-tokenised ids (synthetic prompts): int sequence / tensor [T]
            text_tokens = torch.as_tensor(text, dtype=torch.int32).reshape(1, -1)
        text_tokens = F.pad(text_tokens.to(dev), (0, 1))  # api.py:391
        if text_tokens.shape[-1] >= 400:  # api.py:392
            raise ValueError("Too much text provided. Break the text up into separate segments and re-try inference.")
        auto_conds = None  # the voice clips' mels: what CVVP compares the candidates with (api.py:393-395)
        if voice_samples is not None:
            auto_conditioning, diffusion_conditioning, auto_conds, _ = self.get_conditioning_latents(voice_samples, return_mels=True)
        elif conditioning_latents is not None:
            auto_conditioning, diffusion_conditioning = conditioning_latents
        else:
            auto_conditioning, diffusion_conditioning = self.get_random_conditioning_latents()
        if cvvp_amount > 0:
            if cvvp_amount == 1 and auto_conds is None:
                raise ValueError("cvvp_amount=1 ranks the candidates by CVVP alone, which compares them with the voice's conditioning clips: "
                                 "pass voice_samples (with latents only the reference has nothing to rank by, api.py:462-472)")
            self.load_cvvp()  # api.py:450-453 (loaded even when there are no clips to use it on)
        auto_conditioning = auto_conditioning.to(dev).float()
        diffusion_conditioning = diffusion_conditioning.to(dev).float()
        if max_mel_tokens > self.max_mel_tokens_cap:
            raise ValueError(f"max_mel_tokens={max_mel_tokens} exceeds the capacity this engine was built with "
                             f"(TextToSpeech(max_mel_tokens={self.max_mel_tokens_cap}))")
        sched = Schedule(diffusion_iterations, self.diff_cfg.trained_steps, cond_free, cond_free_k)

        # ---- stage 1: this rank's share of the candidates (api.py:407-427)
        N = int(num_autoregressive_samples)
        lo, hi = tdist.shard_range(N, self.rank, self.world)
        stop = self.ar_cfg.stop_mel_token
        exp_noise = noise.get("exp_noise")
        batches = []
        pre = noise.get("_ar_samples")  # tts_many: this utterance's candidates were decoded in a shared batch already
        if pre is not None:
            batches.append(pre.to(dev).long())
        for c0 in ([] if pre is not None else range(lo, hi, self.autoregressive_batch_size)):
            B = min(self.autoregressive_batch_size, hi - c0)
            self.ar.prefill(auto_conditioning, text_tokens)
            en = exp_noise[:, c0 - lo:c0 - lo + B] if exp_noise is not None else None
            codes, n = self.ar.generate(B, max_mel_tokens, temperature=temperature, top_p=top_p, repetition_penalty=repetition_penalty,
                                        top_k=top_k, seed=seed, row_offset=c0, exp_noise=en, typical_mass=typical_mass)
            batches.append(F.pad(codes, (0, max_mel_tokens - codes.shape[1]), value=stop))  # api.py:425-426
        samples = torch.cat(batches, dim=0)
        ev.mark(1)

        # ---- CLVP ranking (api.py:447-477) + the one collective of the path
        fixed = fix_autoregressive_output(samples, stop)
        # api.py:462-472: CLVP unless cvvp_amount == 1; CVVP (mean over the voice's conditioning clips) when there are clips and cvvp_amount > 0
        scores = self.clvp.score(text_tokens, fixed) if cvvp_amount != 1 else None
        if auto_conds is not None and cvvp_amount > 0:
            cvvp = self.cvvp.score(auto_conds, fixed)
            scores = cvvp if cvvp_amount == 1 else cvvp * cvvp_amount + scores * (1 - cvvp_amount)
        scores_all, codes_all = tdist.gather_candidates(scores, fixed.to(torch.int32)) if self.world > 1 else (scores, fixed.to(torch.int32))
        best = tdist.topk_lowest_index(scores_all, k)
        best_results = codes_all[best].long()
        self.last_best_codes = best_results  # the k ranked winners' codes (tests, sharding checks)
        ev.mark(2)

        # ---- AR latent re-pass for the winners (api.py:516-524)
        best_latents = self.ar.latents(auto_conditioning, text_tokens, best_results)
        ev.mark(3)

        # ---- stage 2 + 3 per winner; winners are spread round-robin over the ranks.  A single winner with
        # conditioning_free is rendered by ranks 0 and 1 together: rank r evaluates denoiser row r of every step.
        split = self.split_diffusion and k == 1 and bool(sched.cond_free)
        wavs = {}
        wav_ok = True
        for i in range(k):
            if split:
                if self.rank > 1:
                    continue
            elif i % self.world != self.rank:
                continue
            codes_i = best_results[i]
            latents = best_latents[i:i + 1]
            latents = latents[:, :calm_trim_length(codes_i)]  # api.py:547-556
            M = latents.shape[1]
            S = M * 4 * 24000 // 22050  # api.py:122
            self.diffusion.condition(latents, diffusion_conditioning, S)
            gen = torch.Generator(device=dev).manual_seed(seed + 7919 * (i + 1))
            x_T = noise.get("x_T")
            x_T = (torch.randn(1, 100, S, device=dev, generator=gen) if x_T is None else x_T.to(dev)) * diffusion_temperature
            step_noise = noise.get("step_noise")
            if step_noise is None:
                step_noise = torch.randn(sched.num_timesteps, 1, 100, S, device=dev, generator=gen)
            if split:
                mel = self.diffusion.sample_split(sched, x_T, step_noise, self.rank, tdist.exchange_rows)
                if self.rank != 0:
                    continue  # rank 1 only lends its GPU to the tail; rank 0 holds the same mel and runs the vocoder
            else:
                mel = self.diffusion.sample(sched, x_T, step_noise)
            ev.mark(4)
            z = noise.get("z")
            z = torch.randn(1, self.voc_cfg.noise_dim, S + 10, device=dev, generator=gen) if z is None else z.to(dev)
            audio = self.vocoder.inference(mel, z)
            finite = torch.isfinite(audio).all()  # (a NaN survives the final clamp: the vocoder stage's overflow check)
            wavs[i] = audio.cpu()
            wav_ok = wav_ok and bool(finite)
        if not wavs:  # this rank had no winner to render
            ev.mark(4)
        ev.mark(5)
        ev.synchronize()
        # operand-overflow guards (fp16 stages): counters the stages' own kernels kept, read after the synchronisation above
        tripped = self._tripped_stages(wav_ok)
        if tripped:
            self._demote(tripped)
            if "ar" in tripped and noise.get("_ar_samples") is not None:
                # candidates decoded ahead of this call (tts_many) came from the stage that just overflowed: decode them again
                noise = {k_: v_ for k_, v_ in noise.items() if k_ != "_ar_samples"}
            return self.tts(text, voice_samples=voice_samples, conditioning_latents=conditioning_latents, k=k, verbose=verbose,
                            use_deterministic_seed=seed, return_deterministic_state=return_deterministic_state,
                            num_autoregressive_samples=num_autoregressive_samples, temperature=temperature, length_penalty=length_penalty,
                            repetition_penalty=repetition_penalty, top_p=top_p, max_mel_tokens=max_mel_tokens, cvvp_amount=cvvp_amount,
                            diffusion_iterations=diffusion_iterations, cond_free=cond_free, cond_free_k=cond_free_k,
                            diffusion_temperature=diffusion_temperature, top_k=top_k,
                            **({"typical_sampling": True, "typical_mass": typical_mass} if typical_mass else {}),
                            **({"noise_override": noise} if noise else {}))
        self.timings = {"ar_s": ev.seconds(0, 1), "clvp_s": ev.seconds(1, 2), "latents_s": ev.seconds(2, 3),
                        "diffusion_s": ev.seconds(3, 4), "vocoder_s": ev.seconds(4, 5), "total_s": ev.seconds(0, 5)}
        # Rendered winners go to rank 0 only (the reference returns the audio to ONE caller); other ranks get None entries.
        if self.world > 1:
            wavs = tdist.collect_on_rank0(wavs, k)
        if wavs is None:
            res = None
            return (res, (seed, text, voice_samples, conditioning_latents)) if return_deterministic_state else res
        wav_candidates = [wavs[i] for i in range(k)]
        res = wav_candidates if len(wav_candidates) > 1 else wav_candidates[0]
        if return_deterministic_state:
            return res, (seed, text, voice_samples, conditioning_latents)
        return res

    @torch.no_grad()
    def tts_many(self, texts, voice_samples=None, conditioning_latents=None, use_deterministic_seed=None, verbose=False, **kwargs):
        """Several utterances of one voice in one call - what tortoise/read.py:66-71 does chunk after chunk with the same seed.
        Returns [tts(text, ...) for text in texts] (k = 1: one clip f32 [1, 1, n] per text), computed with the autoregressive stage
        batched over `utterance_batch` utterances at a time: their candidates share one decode batch (own prefix, own Philox key per
        utterance), so every weight matrix streams once per step for all of them and the sampled codes of an utterance are
        bit-identical to rendering it alone.  With utterance_batch > 1 the CLVP ranking of a wave is ONE speech-tower pass over all its
        candidates (every score the bits of scoring the utterance alone) and the denoiser runs in shared, padded passes; the latent re-pass
        and UnivNet (2 ms per utterance) run per utterance as in tts().  Single-rank instances only (long-form reading spreads whole chunks over the ranks, longform.py)."""
        if self.world != 1:
            raise ValueError("tts_many batches utterances on one GPU: build TextToSpeech(candidate_sharding=False)")
        settings = dict(kwargs)
        k = int(settings.pop("k", 1))
        if k != 1:
            raise NotImplementedError("tts_many renders the top-ranked candidate of every utterance (k = 1)")
        if settings.pop("return_deterministic_state", False):
            raise NotImplementedError("tts_many: return_deterministic_state is a per-call option of tts()")
        N = int(settings.get("num_autoregressive_samples", 512))
        max_mel_tokens = int(settings.get("max_mel_tokens", 500))
        hf = {k_: settings[k_] for k_ in list(settings) if k_ not in (
            "num_autoregressive_samples", "temperature", "length_penalty", "repetition_penalty", "top_p", "max_mel_tokens", "cvvp_amount",
            "diffusion_iterations", "cond_free", "cond_free_k", "diffusion_temperature")}
        if set(hf) - {"top_k", "typical_sampling", "typical_mass"} or settings.get("cvvp_amount", 0) != 0 or N > self.autoregressive_batch_size or N % 4 != 0:
            # anything tts() refuses or the grouped decode cannot hold: let tts() handle (or refuse) it, one utterance at a time
            return [self.tts(t, voice_samples=voice_samples, conditioning_latents=conditioning_latents, k=1, verbose=verbose,
                             use_deterministic_seed=use_deterministic_seed, **settings) for t in texts]
        top_k, typical_mass = sampler_kwargs(hf)
        seed = self.deterministic_state(seed=use_deterministic_seed)
        dev = self.device
        toks = []
        for text in texts:
            if self.enable_redaction and isinstance(text, str) and "[" in text and "]" in text:
                raise NotImplementedError("text with [bracketed] passages needs the wav2vec2 aligner (api.py:583-587); see tts()")
            t = torch.IntTensor(self.tokenizer.encode(text)).unsqueeze(0) if isinstance(text, str) else torch.as_tensor(text, dtype=torch.int32).reshape(1, -1)
            t = F.pad(t.to(dev), (0, 1))
            if t.shape[-1] >= 400:
                raise ValueError("Too much text provided. Break the text up into separate segments and re-try inference.")
            toks.append(t)
        if voice_samples is not None:
            conditioning_latents = self.get_conditioning_latents(voice_samples)
        elif conditioning_latents is None:
            conditioning_latents = self.get_random_conditioning_latents()
        auto_conditioning = conditioning_latents[0].to(dev).float()
        if max_mel_tokens > self.max_mel_tokens_cap:
            raise ValueError(f"max_mel_tokens={max_mel_tokens} exceeds the capacity this engine was built with")
        stop = self.ar_cfg.stop_mel_token
        G = self.utterance_batch
        waves = [list(range(w0, min(w0 + G, len(toks)))) for w0 in range(0, len(toks), G)]

        def ar_wave(idx):
            """One shared decode batch: the candidates of the utterances `idx` -> their code tensors [N, max_mel_tokens]."""
            for g, j in enumerate(idx):
                self.ar.prefill_group(g, len(idx), auto_conditioning, toks[j])
            codes, _ = self.ar.generate(N * len(idx), max_mel_tokens, temperature=settings.get("temperature", .8), top_p=settings.get("top_p", .8),
                                        repetition_penalty=settings.get("repetition_penalty", 2.0), top_k=top_k, seed=seed, row_offset=0,
                                        group_seeds=[seed] * len(idx), typical_mass=typical_mass)
            codes = F.pad(codes, (0, max_mel_tokens - codes.shape[1]), value=stop)
            return [codes[g * N:(g + 1) * N] for g in range(len(idx))]

        out, acc = [None] * len(toks), {}
        if not self.batch_diffusion:
            ev = _StageTimer(2)
            ev.mark(0)
            samples = [smp for idx in waves for smp in ar_wave(idx)]
            # The candidates of every utterance are decoded up front; an overflowed fp16 autoregressive stage must be caught HERE - the
            # per-utterance tts() calls below would otherwise render (and, after their own demotion, re-render) from codes the overflowed
            # stage produced.  (ar.generate() synchronises, so the guard counter is current.)
            ar_guard = getattr(self.ar, "guard", None)
            if ar_guard is not None and ar_guard():
                self._demote(["ar"])
                return self.tts_many(texts, conditioning_latents=conditioning_latents, use_deterministic_seed=seed, verbose=verbose, **kwargs)
            ev.mark(1)
            for j, (t, smp) in enumerate(zip(toks, samples)):
                out[j] = self.tts(t[0, :-1], conditioning_latents=conditioning_latents, k=1, verbose=verbose, use_deterministic_seed=seed,
                                  noise_override={"_ar_samples": smp}, **settings)
                for k_, v in self.timings.items():
                    acc[k_] = acc.get(k_, 0.0) + v
            ev.synchronize()
            acc["ar_s"] = acc.get("ar_s", 0.0) + ev.seconds(0, 1)
            acc["total_s"] = acc.get("total_s", 0.0) + ev.seconds(0, 1)
            self.timings = acc
            return out

        diffusion_conditioning = conditioning_latents[1].to(dev).float()
        sched = Schedule(int(settings.get("diffusion_iterations", 100)), self.diff_cfg.trained_steps, settings.get("cond_free", True),
                         settings.get("cond_free_k", 2))

        def prepare(t, fixed, scores):
            """Winner + latent re-pass of one utterance (api.py:477-524) and its diffusion inputs with the noise tts() would draw."""
            best = tdist.topk_lowest_index(scores, 1)
            best_results = fixed.to(torch.int32)[best].long()
            self.last_best_codes = best_results
            best_latents = self.ar.latents(auto_conditioning, t, best_results)
            latents = best_latents[0:1][:, :calm_trim_length(best_results[0])]
            S = latents.shape[1] * 4 * 24000 // 22050
            gen = torch.Generator(device=dev).manual_seed(seed + 7919)
            x_T = torch.randn(1, 100, S, device=dev, generator=gen) * float(settings.get("diffusion_temperature", 1.0))
            step_noise = torch.randn(sched.num_timesteps, 1, 100, S, device=dev, generator=gen)
            z = torch.randn(1, self.voc_cfg.noise_dim, S + 10, device=dev, generator=gen)
            return (latents, diffusion_conditioning, S, x_T, step_noise), z

        def render(idx, items, zs):
            """Diffusion in shared, padded passes (neighbours in length together: least padding) + UnivNet for the utterances `idx`."""
            order = sorted(range(len(idx)), key=lambda k_: items[k_][2])
            mels = [None] * len(idx)
            for w0 in range(0, len(order), G):
                sel = order[w0:w0 + G]
                if len(sel) == 1:
                    lat_, dc_, S_, x_, n_ = items[sel[0]]
                    self.diffusion.condition(lat_, dc_, S_)
                    mels[sel[0]] = self.diffusion.sample(sched, x_, n_)
                else:
                    for k_, mel in zip(sel, self.diffusion.sample_many(sched, [items[k_] for k_ in sel])):
                        mels[k_] = mel
            for k_, j in enumerate(idx):
                out[j] = self.vocoder.inference(mels[k_], zs[k_]).cpu()

        import time as _time
        on_gpu = torch.device(dev).type == "cuda"  # (the CPU stand-ins of the tests drive the same schedule without streams)

        def sync_current():
            if on_gpu:
                torch.cuda.current_stream().synchronize()

        t_host = {"ar_s": 0.0, "rank_s": 0.0, "render_s": 0.0}
        t_all = _time.perf_counter()
        for idx in waves:
            t0 = _time.perf_counter()
            samples = ar_wave(idx)
            sync_current()
            t_host["ar_s"] += _time.perf_counter() - t0
            t0 = _time.perf_counter()
            # CLVP ranking of the whole wave in ONE speech-tower pass (api.py:460-477 per utterance; read.py:66-71 one call per chunk)
            fixed = [fix_autoregressive_output(smp.to(dev).long(), stop) for smp in samples]
            scores_all = self.clvp.score_groups([toks[j] for j in idx], torch.cat(fixed, dim=0))
            prepared = [prepare(toks[j], fx, scores_all[g * N:(g + 1) * N]) for g, (j, fx) in enumerate(zip(idx, fixed))]
            sync_current()
            t_host["rank_s"] += _time.perf_counter() - t0
            t0 = _time.perf_counter()
            render(idx, [p_[0] for p_ in prepared], [p_[1] for p_ in prepared])
            t_host["render_s"] += _time.perf_counter() - t0
        if on_gpu:
            torch.cuda.synchronize()
        tripped = self._tripped_stages(all(bool(torch.isfinite(o).all()) for o in out))
        if tripped:  # an fp16 stage overflowed: rebuild it with bf16 operands and render the texts again (same seed)
            self._demote(tripped)
            return self.tts_many(texts, conditioning_latents=conditioning_latents, use_deterministic_seed=seed, verbose=verbose, **kwargs)
        # host-clock stage sums
        acc = {"ar_s": t_host["ar_s"], "clvp_s": t_host["rank_s"], "latents_s": 0.0, "diffusion_s": t_host["render_s"], "vocoder_s": 0.0,
               "total_s": _time.perf_counter() - t_all}
        self.timings = acc
        return out

