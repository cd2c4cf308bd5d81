"""TEST INFRASTRUCTURE ONLY — CPU stand-ins for tortoise_tts_amd.stages backed by the ORACLE, so the host logic of the
drop-in class (tortoise_tts_amd/api.py: tokenisation, padding, candidate sharding, fix_autoregressive_output, CLVP top-k,
calm-token trim, return conventions, flag handling) can be driven end to end without a GPU.  The product never imports
this; tests install it with pytest's monkeypatch."""
import time

import numpy as np
import torch

from oracle import tortoise_oracle as O


class FakeTimer:
    def __init__(self, n):
        self.t = [0.0] * n

    def mark(self, i):
        self.t[i] = time.perf_counter()

    def seconds(self, i, j):
        return self.t[j] - self.t[i]

    @staticmethod
    def synchronize():
        pass


class FakeArStage:
    def __init__(self, sd, cfg, device="cpu", dtype=0, max_batch=256, max_text=402, max_new_tokens=500, max_latent_candidates=4,
                 share_weights_with=None, kv_cache=True, max_groups=1):
        self.sd = sd if sd is not None else share_weights_with.sd
        self.dtype = dtype
        self.cfg, self.kv_cache, self.max_batch, self.max_groups = cfg, kv_cache, max_batch, max_groups
        self.groups = None

    def prefill(self, cond_latent, text_tokens):
        self.cond, self.text = cond_latent[:1].float().cpu(), text_tokens[:1].cpu()
        self.groups = None

    def guard(self, reset=True):
        """Operand-overflow guard of the engine stage: with fp16 operands the stand-in reports `FakeArStage.trip` overflows after a
        generation and hands out stop-filled ("cut short") codes, like a decode whose logits went non-finite."""
        return getattr(FakeArStage, "trip", 0) if getattr(self, "dtype", 0) == 1 and getattr(self, "generated", False) else 0

    def close(self):
        self.closed = True

    def prefill_group(self, group, n_groups, cond_latent, text_tokens):
        assert n_groups <= self.max_groups and 0 <= group < n_groups
        if n_groups == 1:
            return self.prefill(cond_latent, text_tokens)
        if group == 0:
            self.groups = [None] * n_groups
        self.groups[group] = (cond_latent[:1].float().cpu(), text_tokens[:1].cpu())
        self.group_batches = getattr(self, "group_batches", 0) + (1 if group == 0 else 0)

    def generate(self, B, max_new, temperature=0.8, top_p=0.8, repetition_penalty=2.0, top_k=50, seed=0, row_offset=0, exp_noise=None,
                 group_seeds=None, typical_mass=0.0):
        assert B <= self.max_batch
        self.generated = True
        if self.dtype == 1 and getattr(FakeArStage, "trip", 0):  # an overflowed fp16 decode: rows cut short with the stop token
            return torch.full((B, max_new), self.cfg.stop_mel_token, dtype=torch.long), max_new
        if self.groups is not None and len(self.groups) > 1:
            # the engine decodes the groups in one batch with per-group prefixes and Philox keys; its contract is "every group's codes
            # equal decoding it alone" (tests/test_gpu_stages.py), which is how the stand-in produces them
            groups, self.groups = self.groups, None
            Bg, outs = B // len(groups), []
            for g, (cond, text) in enumerate(groups):
                self.cond, self.text = cond, text
                outs.append(self.generate(Bg, max_new, temperature, top_p, repetition_penalty, top_k, group_seeds[g] if group_seeds else seed,
                                          row_offset, None, typical_mass=typical_mass)[0])
            n = max(o.shape[1] for o in outs)
            outs = [torch.nn.functional.pad(o, (0, n - o.shape[1]), value=self.cfg.stop_mel_token) for o in outs]
            return torch.cat(outs, dim=0), n
        V = self.cfg.number_mel_codes
        if exp_noise is None:  # one generator per GLOBAL candidate: sharding-invariant like the engine's Philox streams
            rows = []
            for r in range(B):
                g = torch.Generator().manual_seed(int(seed) * 1000003 + row_offset + r)
                rows.append(torch.empty(max_new, V).exponential_(1, generator=g))
            exp_noise = torch.stack(rows, dim=1)
        codes = O.ar_sample_loop(self.sd, self.cfg, self.cond, self.text, B, max_new, exp_noise.cpu(), repetition_penalty, temperature,
                                 top_k, top_p, kv_cache=self.kv_cache, typical_mass=typical_mass or None)
        self.gen_codes = codes
        return codes, codes.shape[1]

    def stream_latents(self, B, n):
        """The latents the engine's decode steps file (tt_ar_stream_latents): by the pinned identity (oracle.ar_latents docstring) they
        are one teacher-forced pass over the sampled codes with the position rule of this handle's kv_cache setting."""
        self.latent_calls = getattr(self, "latent_calls", []) + ["steps"]
        return O.ar_latents(self.sd, self.cfg, self.cond.expand(B, -1), self.text.expand(B, -1), self.gen_codes[:B, :n], stream_positions=self.kv_cache)

    def latents(self, cond_latent, text_tokens, codes, stream_positions=False):
        k = codes.shape[0]
        self.latent_calls = getattr(self, "latent_calls", []) + [bool(stream_positions)]
        return O.ar_latents(self.sd, self.cfg, cond_latent.float().cpu().expand(k, -1), text_tokens.cpu().expand(k, -1), codes.cpu(),
                            stream_positions=stream_positions)

    def generate_stream(self, B, max_new, chunk, first_chunk=None, **kw):
        """The resumable loop of the streaming path: the oracle loop is not resumable, so the whole sequence is sampled once and
        handed out in the same pieces the engine would produce (the engine test checks chunks == one-shot bit for bit)."""
        codes, n = self.generate(B, max_new, **kw)
        done_at = n if n < max_new or bool((codes[:, -1] == self.cfg.stop_mel_token).all()) else max_new
        pos, first = 0, True
        while pos < done_at:
            pos = min(pos + ((first_chunk or chunk) if first else chunk), done_at)
            first = False
            yield codes[:, :pos], pos >= done_at


class FakeHifiganStage:
    def __init__(self, sd_folded, cfg, device="cpu", dtype=0, max_latents=512):
        self.sd, self.cfg = sd_folded, cfg

    def inference(self, latents, g):
        return O.hifigan_inference(self.sd, self.cfg, latents.float().cpu(), g.float().cpu().reshape(1, -1))


class FakeClvpStage:
    def __init__(self, sd, cfg, device="cpu", dtype=0, max_rows=0):
        self.sd, self.cfg = sd, cfg

    def score(self, text_tokens, codes):
        B = codes.shape[0]
        return O.clvp_score(self.sd, self.cfg, text_tokens[:1].long().cpu().repeat(B, 1), codes.long().cpu())

    def score_groups(self, texts, codes):
        """The engine scores the utterances of a wave in one speech-tower pass with every score equal to score() alone
        (tests/test_gpu_r6.py): the stand-in scores them alone."""
        N = codes.shape[0] // len(texts)
        self.grouped = getattr(self, "grouped", []) + [len(texts)]
        return torch.cat([self.score(t.reshape(1, -1), codes[g * N:(g + 1) * N]) for g, t in enumerate(texts)])


class FakeCvvpStage:
    def __init__(self, sd, cfg, device="cpu", dtype=0, max_rows=0, max_cond_frames=520):
        self.sd, self.cfg, self.dtype, self.calls = sd, cfg, dtype, 0

    def score(self, auto_conds, codes):
        self.calls += 1
        return O.cvvp_score(self.sd, self.cfg, auto_conds.float().cpu(), codes.long().cpu())

    def close(self):
        pass


class FakeDiffusionStage:
    def __init__(self, sd, cfg, device="cpu", dtype=0, max_seq=0, max_codes=0, max_steps=0, max_batch=1):
        self.sd, self.cfg, self.max_batch, self.dtype = sd, cfg, max_batch, dtype

    def sample_many(self, sched, items):
        """The engine pushes the utterances through shared denoiser passes with every one treated as if alone (tests/test_gpu_parity_r3.py):
        the stand-in renders them alone."""
        assert 1 <= len(items) <= self.max_batch
        self.batched = getattr(self, "batched", []) + [len(items)]
        out = []
        for lat, cond, S, x_T, noise in items:
            self.condition(lat, cond, S)
            out.append(self.sample(sched, x_T, noise))
        return out

    def condition(self, latents, cond_latent, S):
        self.S = S
        self.emb = O.diffusion_timestep_independent(self.sd, self.cfg, latents.float().cpu(), cond_latent.float().cpu(), S)

    def guard(self, reset=True):
        """Operand-overflow guard of the engine stage: the stand-in trips `self.trip` times (tests of the demotion path)."""
        n = getattr(FakeDiffusionStage, "trip", 0) if self.dtype == 1 else 0
        if n and reset:
            FakeDiffusionStage.trip = 0
        return n

    def close(self):
        self.closed = True

    def sample_split(self, sched, x_T, step_noise, row, exchange):
        """The split tail's host protocol (one row exchange per step over the pair group); the stand-in has no per-row denoiser, so
        both participants walk the whole loop and exchange a tag per step - what is exercised is the collective pattern."""
        rows = torch.zeros(2, 4, 3)
        for step in range(sched.num_timesteps):
            exchange(rows, torch.full((4, 3), float(10 * step + row)))
            assert float(rows[0, 0, 0]) == 10.0 * step and float(rows[1, 0, 0]) == 10.0 * step + 1
        self.split_steps = sched.num_timesteps
        return self.sample(sched, x_T, step_noise)

    def sample(self, sched, x_T, step_noise):
        osched = O.Schedule(sched.num_timesteps, self.cfg.trained_steps, sched.cond_free, sched.cond_free_k)
        assert np.array_equal(osched.timestep_map, sched.timestep_map)
        return O.denormalize_tacotron_mel(O.p_sample_loop(self.sd, self.cfg, osched, self.emb, x_T.float().cpu(), step_noise.float().cpu()))


class FakeVocoderStage:
    def __init__(self, sd_folded, cfg, device="cpu", dtype=0, max_frames=0):
        self.sd, self.cfg = sd_folded, cfg

    def inference(self, mel, z):
        return O.univnet_inference(self.sd, self.cfg, mel.float().cpu(), z.float().cpu())


class FakeRandomLatentStage:
    def __init__(self, sd_auto, sd_diffuser, device="cpu", dtype=0):
        self.sds = (sd_auto, sd_diffuser)
        self.channels = (sd_auto["layers.0.weight"].shape[0], sd_diffuser["layers.0.weight"].shape[0])

    def latents(self, r_auto, r_diffuser):
        return O.random_latent_converter(self.sds[0], r_auto), O.random_latent_converter(self.sds[1], r_diffuser)


class FakeConditioningStage:
    def __init__(self, sd_ar, sd_diff, ar_cfg, diff_cfg, device="cpu", dtype=0, max_frames=1024):
        self.sd_ar, self.sd_diff, self.ar_cfg, self.diff_cfg = sd_ar, sd_diff, ar_cfg, diff_cfg

    def auto_latent(self, mels):
        return O.ar_get_conditioning(self.sd_ar, self.ar_cfg, torch.stack([m.float().cpu() for m in mels], dim=1))

    def diffusion_latent(self, mels):
        return O.diffusion_get_conditioning(self.sd_diff, self.diff_cfg, torch.stack([m.float().cpu() for m in mels], dim=1))


def install(monkeypatch):
    """Route tortoise_tts_amd.api onto the CPU stand-ins (and a CPU 'device')."""
    from tortoise_tts_amd import api
    monkeypatch.setattr(api.stages, "ArStage", FakeArStage)
    monkeypatch.setattr(api.stages, "ClvpStage", FakeClvpStage)
    monkeypatch.setattr(api.stages, "CvvpStage", FakeCvvpStage)
    monkeypatch.setattr(api.stages, "DiffusionStage", FakeDiffusionStage)
    monkeypatch.setattr(api.stages, "VocoderStage", FakeVocoderStage)
    monkeypatch.setattr(api.stages, "RandomLatentStage", FakeRandomLatentStage)
    monkeypatch.setattr(api.stages, "ConditioningStage", FakeConditioningStage)
    monkeypatch.setattr(api.stages, "HifiganStage", FakeHifiganStage)
    monkeypatch.setattr(api.E, "require_gpu", lambda device=None: torch.device("cpu"))
    monkeypatch.setattr(api, "_StageTimer", FakeTimer)
