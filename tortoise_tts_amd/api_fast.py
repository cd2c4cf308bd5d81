"""Drop-in for tortoise.api_fast.TextToSpeech on the MI355X engine: the streaming / low-latency path
(reference tortoise/api_fast.py:180-540).

One autoregressive sample, no CLVP, no diffusion, no UnivNet: GPT latents go straight into the HiFi-GAN decoder
(hifigan_decoder.py:159-294).  The engine pieces are the AR stage of the main path (prefill, hipGraph decode loop,
teacher-forced latent re-pass) and csrc/hifigan.hip.

  * tts(text, ...)          api_fast.py:421-519: inference_speech (1 sequence) -> autoregressive(..., return_latent=True)
                            -> hifi_decoder.inference(latents, auto_conditioning).  Returns wav f32 [1, 1, S] on the CPU.
  * tts_stream(text, ...)   api_fast.py:311-420: a generator of waveform chunks.  The reference pulls (token, latent) pairs out
                            of HF's sampling loop and, every `stream_chunk_size` tokens (first chunk: 60), decodes ALL latents so
                            far and cross-fades the new part in (handle_chunks).  Here the decode loop is resumed chunk by chunk
                            on the device (tt_ar_generate_chunk) and every decode step files its own latent
                            (tt_ar_stream_latents), as the reference's loop does.  Those per-step states equal one teacher-forced
                            pass over the codes - plain mel positions with kv_cache=False (the default), the cached decode's
                            0, 2, 3, ... with kv_cache=True (autoregressive.py:134-149; oracle.ar_latents(stream_positions=True),
                            pinned live against the reference's sample_stream) - which the tests use as the check.
  * handle_chunks           api_fast.py:275-309, restated (host-side tensor slicing / cross-fade).

Sampling noise comes from the engine's Philox streams keyed by use_deterministic_seed (seeds are not portable between
generators: parity is "same latents -> same waveform", tests/test_gpu_stages.py::test_hifigan_decoder).
"""
import os
import random

import torch
import torch.nn.functional as F

from . import engine as E
from . import stages
from . import weights as W
from .api import MODELS_DIR, _load_state_dict, _load_file, sampler_kwargs
from .config import ARConfig, HifiganConfig


class TextToSpeech:
    """api_fast.py:180-229.  Engine-only keyword arguments as in tortoise_tts_amd.api.TextToSpeech: state_dicts
    ('autoregressive', 'hifidecoder', 'rlg_auto'), dtype, configs ('ar', 'hifigan'), max_mel_tokens."""

    def __init__(self, autoregressive_batch_size=None, models_dir=MODELS_DIR, enable_redaction=True, kv_cache=False,
                 use_deepspeed=False, half=False, device=None, tokenizer_vocab_file=None, tokenizer_basic=False, *,
                 state_dicts=None, dtype=None, configs=None, max_mel_tokens=500, max_text_tokens=402):
        self.models_dir = models_dir
        if use_deepspeed:
            raise NotImplementedError("use_deepspeed: DeepSpeed kernel injection is a CUDA-only reference option")
        self.enable_redaction = bool(enable_redaction)
        self.kv_cache = bool(kv_cache)
        self.half = bool(half)
        self.device = E.require_gpu(device)
        if dtype is None:
            dtype = "fp16" if half else "bf16"
        elif half and dtype not in ("fp16", "f16"):
            raise ValueError(f"half=True asks for fp16 operands but dtype={dtype!r} was also given")
        self.dtype = {"bf16": E.TT_BF16, "fp16": E.TT_F16, "f16": E.TT_F16}[dtype]
        cfgs = configs or {}
        self.ar_cfg = cfgs.get("ar", ARConfig())
        self.hifi_cfg = cfgs.get("hifigan", HifiganConfig())
        self._state_dicts = state_dicts or {}
        self.autoregressive_batch_size = 1
        self.tokenizer_args = (tokenizer_vocab_file, tokenizer_basic)
        self._tokenizer = None
        self.max_mel_tokens_cap = max_mel_tokens
        self.ar = stages.ArStage(self._sd("autoregressive"), self.ar_cfg, self.device, self.dtype, max_batch=1, max_text=max_text_tokens,
                                 max_new_tokens=max_mel_tokens, max_latent_candidates=1, kv_cache=self.kv_cache)
        hsd = self._sd("hifidecoder")
        if any(k.endswith("weight_v") for k in hsd):
            hsd = W.fold_weight_norm(hsd)
        self.hifi_decoder = stages.HifiganStage(hsd, self.hifi_cfg, self.device, self.dtype, max_latents=max_mel_tokens + 8)
        self.rlg_auto = None
        self.conditioning = None
        self.mel_front_end = None
        self.stream_latents_from = "steps"
        self.stop_mel_token = self.ar_cfg.stop_mel_token
        self.mel_length_compression = self.ar_cfg.mel_length_compression

    def _sd(self, name):
        if name in self._state_dicts:
            return self._state_dicts[name]
        if name == "hifidecoder":
            return _load_file(self.models_dir, "hifidecoder.pth")
        return _load_state_dict(self.models_dir, name)

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from .text import VoiceBpeTokenizer
            self._tokenizer = VoiceBpeTokenizer(self.tokenizer_args[0], self.tokenizer_args[1], self.models_dir)
        return self._tokenizer

    # ------------------------------------------------------------------ conditioning (api_fast.py:230-260)
    def get_conditioning_latents(self, voice_samples, return_mels=False):
        """Only the autoregressive latent exists on this path (there is no diffusion stage).  voice_samples: 22.05 kHz clips
        or ready auto mels f32 [1, 80, T] (see tortoise_tts_amd.api.TextToSpeech.get_conditioning_latents)."""
        if torch.is_tensor(voice_samples):
            voice_samples = [voice_samples]
        if self.conditioning is None:
            self.conditioning = stages.ConditioningStage(self._sd("autoregressive"), None, self.ar_cfg, None, self.device, self.dtype)
        mels = []
        for vs in voice_samples:
            if torch.is_tensor(vs) and vs.dim() >= 2 and vs.shape[-2] == 80:
                mels.append(vs.reshape(1, 80, vs.shape[-1]))
            else:
                if self.mel_front_end is None:
                    from .audio import MelFrontEnd
                    self.mel_front_end = MelFrontEnd(self.models_dir)
                mels.append(self.mel_front_end.auto_mel(vs.to(self.device)))
        return self.conditioning.auto_latent(mels)

    def get_random_conditioning_latents(self):
        if self.rlg_auto is None:
            sd = self._sd("rlg_auto")
            self.rlg_auto = stages.RandomLatentStage(sd, sd, self.device, self.dtype)
        ca = self.rlg_auto.channels[0]
        r = torch.randn(1, ca)
        return self.rlg_auto.latents(r, r)[0]

    def deterministic_state(self, seed=None):
        seed = int(torch.seed() % (2 ** 31)) if seed is None else int(seed)
        torch.manual_seed(seed)
        random.seed(seed)
        return seed

    def _prepare(self, text, voice_samples, conditioning_latents, max_mel_tokens):
        if isinstance(text, str):
            ids = self.tokenizer.encode(text)
        else:
            ids = [int(t) for t in text]
        text_tokens = F.pad(torch.tensor(ids, dtype=torch.int32, device=self.device)[None], (0, 1))
        if text_tokens.shape[-1] >= 400:
            raise ValueError("Too much text provided. Break the text up into separate segments and re-try inference.")  # api_fast.py:371
        if not 1 <= max_mel_tokens <= self.max_mel_tokens_cap:
            raise ValueError(f"max_mel_tokens={max_mel_tokens} outside [1, {self.max_mel_tokens_cap}] (engine capacity)")
        if voice_samples is not None:
            cond = self.get_conditioning_latents(voice_samples)
        elif conditioning_latents is not None:
            cond = conditioning_latents[0] if isinstance(conditioning_latents, (tuple, list)) else conditioning_latents
        else:
            cond = self.get_random_conditioning_latents()
        return text_tokens, cond.to(self.device).float().reshape(1, -1)

    @staticmethod
    def _check_kwargs(k, cvvp_amount, hf_generate_kwargs):
        """Same refusals as tortoise_tts_amd.api.TextToSpeech.tts: a sampling option the on-device sampler cannot honour raises
        instead of being dropped.  `k` and `cvvp_amount` are accepted and unused exactly as in the reference, whose fast path always
        decodes ONE autoregressive sample into one clip - there is no candidate ranking CLVP or CVVP could take part in
        (api_fast.py:316, 426, 421-519).  Returns (top_k, typical_mass) (api.sampler_kwargs)."""
        return sampler_kwargs(hf_generate_kwargs)

    # ------------------------------------------------------------------ non-streaming (api_fast.py:421-519)
    @torch.no_grad()
    def tts(self, text, voice_samples=None, k=1, verbose=True, use_deterministic_seed=None, conditioning_latents=None,
            num_autoregressive_samples=512, temperature=.8, length_penalty=1, repetition_penalty=2.0, top_p=.8, max_mel_tokens=500,
            cvvp_amount=.0, **hf_generate_kwargs):
        top_k, typical_mass = self._check_kwargs(k, cvvp_amount, hf_generate_kwargs)
        seed = self.deterministic_state(seed=use_deterministic_seed)
        text_tokens, cond = self._prepare(text, voice_samples, conditioning_latents, max_mel_tokens)
        self.ar.prefill(cond, text_tokens)
        codes, _ = self.ar.generate(1, max_mel_tokens, temperature=temperature, top_p=top_p, repetition_penalty=float(repetition_penalty),
                                    top_k=top_k, seed=seed, row_offset=0, typical_mass=typical_mass)
        self.last_codes = codes
        latents = self.ar.latents(cond, text_tokens, codes)          # api_fast.py:510-514 (return_latent=True)
        wav = self.hifi_decoder.inference(latents, cond)             # api_fast.py:517
        return wav.cpu()

    def tts_with_preset(self, text, preset="fast", **kwargs):
        """api_fast.py:262-273: the preset table of this class only feeds kwargs to tts() (diffusion settings are unused on this
        path); a generator over the result, like the reference."""
        settings = {"temperature": .8, "length_penalty": 1.0, "repetition_penalty": 2.0, "top_p": .8}
        presets = {"ultra_fast": {"num_autoregressive_samples": 1}, "fast": {"num_autoregressive_samples": 32},
                   "standard": {"num_autoregressive_samples": 256}, "high_quality": {"num_autoregressive_samples": 256}}
        settings.update(presets[preset])
        settings.update({k_: v for k_, v in kwargs.items() if k_ not in ("cond_free_k", "diffusion_temperature", "diffusion_iterations", "cond_free")})
        for audio_frame in self.tts(text, **settings):
            yield audio_frame

    # ------------------------------------------------------------------ streaming (api_fast.py:275-420)
    @staticmethod
    def handle_chunks(wav_gen, wav_gen_prev, wav_overlap, overlap_len):
        """api_fast.py:275-309: the part of the newly decoded waveform that has not been emitted yet, cross-faded over the overlap."""
        wav_chunk = wav_gen[:-overlap_len]
        if wav_gen_prev is not None:
            wav_chunk = wav_gen[(wav_gen_prev.shape[0] - overlap_len):-overlap_len]
        if wav_overlap is not None:
            if overlap_len > len(wav_chunk):
                if wav_gen_prev is not None:
                    wav_chunk = wav_gen[(wav_gen_prev.shape[0] - overlap_len):]
                else:
                    wav_chunk = wav_gen[-overlap_len:]
                return wav_chunk, wav_gen, None
            wav_chunk = wav_chunk.clone()
            fade_in = torch.linspace(0.0, 1.0, overlap_len, device=wav_chunk.device)
            crossfade = wav_chunk[:overlap_len] * fade_in
            wav_chunk[:overlap_len] = wav_overlap * torch.linspace(1.0, 0.0, overlap_len, device=wav_overlap.device)
            wav_chunk[:overlap_len] += crossfade
        wav_overlap = wav_gen[-overlap_len:]
        return wav_chunk, wav_gen, wav_overlap

    def _stream_latents(self, cond, text_tokens, codes):
        """The latent half of the (token, latent) pairs of the reference's generator (api_fast.py:405-414): filed by the decode
        steps themselves (tt_ar_stream_latents), exactly where the reference takes them from.  `stream_latents_from="pass"` keeps
        the earlier formulation - one teacher-forced pass over the codes so far (plain positions for kv_cache=False, the cached
        decode's 0, 2, 3, ... for kv_cache=True) - which the tests hold equal to the per-step latents within the operand tolerance."""
        if self.stream_latents_from == "steps":
            return self.ar.stream_latents(1, codes.shape[1])
        return self.ar.latents(cond, text_tokens, codes, stream_positions=self.kv_cache)

    @torch.no_grad()
    def tts_stream(self, text, voice_samples=None, conditioning_latents=None, k=1, verbose=True, use_deterministic_seed=None,
                   return_deterministic_state=False, overlap_wav_len=1024, stream_chunk_size=40,
                   num_autoregressive_samples=512, temperature=.8, length_penalty=1, repetition_penalty=2.0, top_p=.8, max_mel_tokens=500,
                   cvvp_amount=.0, diffusion_iterations=100, cond_free=True, cond_free_k=2, diffusion_temperature=1.0,
                   **hf_generate_kwargs):
        top_k, typical_mass = self._check_kwargs(k, cvvp_amount, hf_generate_kwargs)
        seed = self.deterministic_state(seed=use_deterministic_seed)
        text_tokens, cond = self._prepare(text, voice_samples, conditioning_latents, max_mel_tokens)
        self.ar.prefill(cond, text_tokens)
        chunk = stream_chunk_size if stream_chunk_size > 0 else max_mel_tokens
        first = max(chunk, 60) if stream_chunk_size > 0 else max_mel_tokens  # first_buffer = 60 (api_fast.py:401, 412)
        wav_gen_prev, wav_overlap = None, None
        emitted = 0          # (token, latent) pairs already decoded into an emitted chunk
        threshold = first    # pairs the reference buffers before the next decode (api_fast.py:412)
        for codes, done in self.ar.generate_stream(1, max_mel_tokens, chunk, first_chunk=first, temperature=temperature, top_p=top_p,
                                                   repetition_penalty=float(repetition_penalty), top_k=top_k, seed=seed,
                                                   typical_mass=typical_mass):
            if done and codes.shape[1] > 0 and int(codes[0, -1]) == self.stop_mel_token:
                codes = codes[:, :-1]  # the reference's generator stops BEFORE yielding the stop token's pair
            if codes.shape[1] == 0:
                break
            self.last_codes = codes
            latents = self._stream_latents(cond, text_tokens, codes)
            wav_gen = self.hifi_decoder.inference(latents, cond).reshape(-1)
            wav_chunk, wav_gen_prev, wav_overlap = self.handle_chunks(wav_gen, wav_gen_prev, wav_overlap, overlap_wav_len)
            yield wav_chunk
            if done:
                # The reference decodes once per filled buffer AND once more when its generator raises StopIteration
                # (api_fast.py:405-420).  When the sequence ends exactly on a buffer boundary that last pass sees the same latents
                # again and handle_chunks hands out the withheld overlap tail: mirror it.
                if stream_chunk_size > 0 and codes.shape[1] - emitted == threshold:
                    wav_chunk, wav_gen_prev, wav_overlap = self.handle_chunks(wav_gen, wav_gen_prev, wav_overlap, overlap_wav_len)
                    yield wav_chunk
                break
            emitted = codes.shape[1]
            threshold = chunk
