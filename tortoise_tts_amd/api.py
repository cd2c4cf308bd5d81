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
Out of scope this round (raise, never silently fall back): voice_samples -> conditioning latents
(SURVEY.md §8f-3), CVVP (cvvp_amount != 0, removed upstream), wav2vec redaction, DeepSpeed flag.
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
from .config import ARConfig, CLVPConfig, DiffusionConfig, VocoderConfig, PRESETS, BASE_SETTINGS, CALM_TOKEN
from .schedule import Schedule

MODELS_DIR = os.environ.get("TORTOISE_MODELS_DIR", os.path.join(os.path.expanduser("~"), ".cache", "tortoise", "models"))
MODEL_FILES = {"autoregressive": "autoregressive.pth", "clvp": "clvp2.pth", "diffusion": "diffusion_decoder.pth",
               "vocoder": "vocoder.pth"}


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


class TextToSpeech:
    """Main entry point; see the module docstring.  Engine-only keyword arguments (all optional, after
    the reference's): `state_dicts` (dict of reference-layout state_dicts instead of files in
    models_dir), `dtype` ('bf16' | 'fp16' MFMA operand type), `max_candidates` (per-GPU decode batch
    capacity), `configs` (ARConfig/CLVPConfig/DiffusionConfig/VocoderConfig overrides for tests)."""

    def __init__(self, autoregressive_batch_size=None, models_dir=MODELS_DIR, enable_redaction=True, kv_cache=False,
                 use_deepspeed=False, half=False, device=None, tokenizer_vocab_file=None, tokenizer_basic=False, *,
                 state_dicts=None, dtype="bf16", max_candidates=256, configs=None, max_mel_tokens=500, max_text_tokens=402,
                 decode_streams=1):
        self.models_dir = models_dir
        if use_deepspeed:
            raise NotImplementedError("use_deepspeed: DeepSpeed kernel injection is a CUDA-only reference option; the MI355X engine "
                                      "always runs its own fused HIP path")
        self.enable_redaction = False  # wav2vec2 redaction is outside the hot path
        self.rank, self.world = tdist.world()
        # With >= 2 ranks a single winner's diffusion tail is split over ranks 0 and 1 (conditioned / conditioning-free
        # row each, one exchange per step).  TT_SPLIT_DIFFUSION=0 keeps the whole tail on rank 0.
        self.split_diffusion = self.world >= 2 and os.environ.get("TT_SPLIT_DIFFUSION", "1") != "0"
        if self.split_diffusion:
            tdist.pair_group()  # collective over all ranks: create it once, here, where every rank passes
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if device is None or torch.device(device).type != "cuda":
            raise E.EngineError("TextToSpeech needs an MI355X (gfx950) device; the engine has no CPU path")
        self.device = torch.device(device)
        self.dtype = {"bf16": E.TT_BF16, "fp16": E.TT_F16, "f16": E.TT_F16}[dtype]
        cfgs = configs or {}
        self.ar_cfg = cfgs.get("ar", ARConfig())
        self.clvp_cfg = cfgs.get("clvp", CLVPConfig())
        self.diff_cfg = cfgs.get("diffusion", DiffusionConfig())
        self.voc_cfg = cfgs.get("vocoder", VocoderConfig())
        sds = state_dicts or {}

        def sd(name):
            return sds[name] if name in sds else _load_state_dict(models_dir, name)

        # the reference's default AR batch is 16 on a >=14 GB GPU (api.py:148-172); 288 GB of HBM3E decodes
        # every candidate of this rank in one batch unless the caller asks for smaller batches.
        self.autoregressive_batch_size = int(autoregressive_batch_size or max_candidates)
        cap = min(self.autoregressive_batch_size, max_candidates)
        self.tokenizer_args = (tokenizer_vocab_file, tokenizer_basic)
        self._tokenizer = None
        self.max_mel_tokens_cap = max_mel_tokens
        max_S = max_mel_tokens * 4 * 24000 // 22050 + 8
        self.ar = stages.ArStage(sd("autoregressive"), self.ar_cfg, self.device, self.dtype, max_batch=cap,
                                 max_text=max_text_tokens, max_new_tokens=max_mel_tokens, max_latent_candidates=4)
        # Optional: decode several candidate sub-batches concurrently on their own streams (shared weights; sampled
        # codes do not change because Philox streams are keyed by the global candidate index).  Measured on MI355X:
        # no gain (1 stream 0.505 s, 2 streams 0.515 s, 4 streams 0.915 s for 256 x 200 tokens) - the decode GEMMs
        # are weight-streaming bound, so splitting the batch only streams the weights more often.  Default 1.
        self.decode_streams = max(1, int(decode_streams))
        self.ar_extra = [stages.ArStage(None, self.ar_cfg, self.device, self.dtype, max_batch=-(-cap // self.decode_streams),
                                        max_text=max_text_tokens, max_new_tokens=max_mel_tokens, max_latent_candidates=1,
                                        share_weights_with=self.ar) for _ in range(self.decode_streams - 1)]
        self.clvp = stages.ClvpStage(sd("clvp"), self.clvp_cfg, self.device, self.dtype, max_rows=max(cap, 8) * max_mel_tokens)
        self.diffusion = stages.DiffusionStage(sd("diffusion"), self.diff_cfg, self.device, self.dtype, max_seq=max_S,
                                               max_codes=max_mel_tokens + 8, max_steps=512)
        voc_sd = sd("vocoder")
        if any(k.endswith("weight_v") for k in voc_sd):
            voc_sd = W.fold_weight_norm(voc_sd)  # UnivNetGenerator.eval(inference=True), vocoder.py:284-298
        self.vocoder = stages.VocoderStage(voc_sd, self.voc_cfg, self.device, self.dtype, max_frames=max_S)
        # attributes the reference exposes and callers touch (api.py:408, 523)
        self.stop_mel_token = self.ar_cfg.stop_mel_token
        self.mel_length_compression = self.ar_cfg.mel_length_compression
        self.timings = {}

    # ------------------------------------------------------------------ reference helpers
    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from .text import VoiceBpeTokenizer
            self._tokenizer = VoiceBpeTokenizer(self.tokenizer_args[0], self.tokenizer_args[1], self.models_dir)
        return self._tokenizer

    def get_conditioning_latents(self, voice_samples, return_mels=False):
        raise NotImplementedError("voice_samples -> conditioning latents (ConditioningEncoder / contextual_embedder / STFT front-end, "
                                  "api.py:258-299) is not on the accelerated path yet (SURVEY.md §8f-3); pass conditioning_latents= "
                                  "(e.g. the .pth latent files the reference caches per voice)")

    def get_random_conditioning_latents(self):
        raise NotImplementedError("random-voice latents need the reference's rlg_auto.pth / rlg_diffuser.pth MLPs (api.py:301-309), "
                                  "which are outside the hot path; pass conditioning_latents=")

    def deterministic_state(self, seed=None):
        """api.py:598-609."""
        seed = int(torch.seed() % (2 ** 31)) if seed is None else int(seed)
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
        top_k = int(hf_generate_kwargs.pop("top_k", 50))  # HF GenerationConfig default the reference inherits
        if hf_generate_kwargs:
            raise NotImplementedError(f"unsupported generate kwargs {sorted(hf_generate_kwargs)}: the on-device sampler implements "
                                      f"temperature / top_k / top_p / repetition_penalty (length_penalty is a no-op when sampling)")
        if cvvp_amount != 0:
            raise NotImplementedError("CVVP was removed upstream (CHANGELOG) and is not part of the accelerated path")
        dev = self.device
        seed = self.deterministic_state(seed=use_deterministic_seed)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record()

        if isinstance(text, str):
            text_tokens = torch.IntTensor(self.tokenizer.encode(text)).unsqueeze(0)
        else:  # pre-tokenised ids (synthetic prompts): int sequence / tensor [T]
            text_tokens = torch.as_tensor(text, dtype=torch.int32).reshape(1, -1)
        text_tokens = F.pad(text_tokens.to(dev), (0, 1))  # api.py:391
        assert text_tokens.shape[-1] < 400, "Too much text provided. Break the text up into separate segments and re-try inference."
        if voice_samples is not None:
            auto_conditioning, diffusion_conditioning = self.get_conditioning_latents(voice_samples)
        elif conditioning_latents is not None:
            auto_conditioning, diffusion_conditioning = conditioning_latents
        else:
            auto_conditioning, diffusion_conditioning = self.get_random_conditioning_latents()
        auto_conditioning = auto_conditioning.to(dev).float()
        diffusion_conditioning = diffusion_conditioning.to(dev).float()
        assert max_mel_tokens <= self.max_mel_tokens_cap
        sched = Schedule(diffusion_iterations, self.diff_cfg.trained_steps, cond_free, cond_free_k)

        # ---- stage 1: this rank's share of the candidates (api.py:407-427)
        N = int(num_autoregressive_samples)
        lo, hi = tdist.shard_range(N, self.rank, self.world)
        stop = self.ar_cfg.stop_mel_token
        exp_noise = noise.get("exp_noise")
        jobs = []  # (first global candidate, count)
        for b0 in range(lo, hi, self.autoregressive_batch_size):
            B = min(self.autoregressive_batch_size, hi - b0)
            nsub = self.decode_streams if B >= 32 * self.decode_streams else 1
            per = -(-B // nsub)
            for j in range(nsub):
                c0 = b0 + j * per
                if c0 < b0 + B:
                    jobs.append((c0, min(per, b0 + B - c0)))

        def decode(stage, c0, B, out, idx, stream):
            with torch.cuda.stream(stream):
                stage.prefill(auto_conditioning, text_tokens)
                en = exp_noise[:, c0 - lo:c0 - lo + B] if exp_noise is not None else None
                codes, n = stage.generate(B, max_mel_tokens, temperature=temperature, top_p=top_p, repetition_penalty=repetition_penalty,
                                          top_k=top_k, seed=seed, row_offset=c0, exp_noise=en)
                out[idx] = F.pad(codes, (0, max_mel_tokens - codes.shape[1]), value=stop)  # api.py:425-426

        batches = [None] * len(jobs)
        handles = [self.ar] + self.ar_extra
        cur = torch.cuda.current_stream()
        for w0 in range(0, len(jobs), len(handles)):
            wave = jobs[w0:w0 + len(handles)]
            if len(wave) == 1:
                decode(handles[0], wave[0][0], wave[0][1], batches, w0, cur)
                continue
            import threading
            streams = [torch.cuda.Stream(device=dev) for _ in wave]
            for st_ in streams:
                st_.wait_stream(cur)
            threads = [threading.Thread(target=decode, args=(handles[i], c0, B, batches, w0 + i, streams[i]))
                       for i, (c0, B) in enumerate(wave)]
            for t_ in threads:
                t_.start()
            for t_ in threads:
                t_.join()
            for st_ in streams:
                cur.wait_stream(st_)
        samples = torch.cat(batches, dim=0)
        ev[1].record()

        # ---- CLVP ranking (api.py:447-477) + the one collective of the path
        fixed = fix_autoregressive_output(samples, stop)
        scores = self.clvp.score(text_tokens, fixed)
        scores_all, codes_all = tdist.gather_candidates(scores, fixed.to(torch.int32))
        best = tdist.topk_lowest_index(scores_all, k)
        best_results = codes_all[best].long()
        self.last_best_codes = best_results  # the k ranked winners' codes (tests, sharding checks)
        ev[2].record()

        # ---- AR latent re-pass for the winners (api.py:516-524)
        best_latents = self.ar.latents(auto_conditioning, text_tokens, best_results)
        ev[3].record()

        # ---- stage 2 + 3 per winner; winners are spread round-robin over the ranks.  A single winner with
        # conditioning_free is rendered by ranks 0 and 1 together: rank r evaluates denoiser row r of every step.
        split = self.split_diffusion and k == 1 and bool(sched.cond_free)
        wavs = {}
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
            ev[4].record()
            z = noise.get("z")
            z = torch.randn(1, self.voc_cfg.noise_dim, S + 10, device=dev, generator=gen) if z is None else z.to(dev)
            wavs[i] = self.vocoder.inference(mel, z).cpu()
        if not wavs:  # this rank had no winner to render
            ev[4].record()
        ev[5].record()
        torch.cuda.synchronize()
        self.timings = {"ar_s": ev[0].elapsed_time(ev[1]) / 1e3, "clvp_s": ev[1].elapsed_time(ev[2]) / 1e3,
                        "latents_s": ev[2].elapsed_time(ev[3]) / 1e3, "diffusion_s": ev[3].elapsed_time(ev[4]) / 1e3,
                        "vocoder_s": ev[4].elapsed_time(ev[5]) / 1e3, "total_s": ev[0].elapsed_time(ev[5]) / 1e3}
        if self.world > 1:
            import torch.distributed as tdd
            gathered = [None] * self.world
            tdd.all_gather_object(gathered, wavs)
            wavs = {i: w for d in gathered for i, w in d.items()}
        wav_candidates = [wavs[i] for i in range(k)]
        res = wav_candidates if len(wav_candidates) > 1 else wav_candidates[0]
        if return_deterministic_state:
            return res, (seed, text, voice_samples, conditioning_latents)
        return res
