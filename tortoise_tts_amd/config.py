"""Hyper-parameters of the four networks on the hot path.

The defaults are the values the reference hard-codes when it builds its models
(reference: tortoise/api.py:217-236).  Smaller instances (fewer layers / narrower) are
used by the parity tests; every kernel is shape-generic within the limits stated in
DESIGN.md (head_dim == 64, channels % 32 == 0).
"""
from dataclasses import dataclass, field
from typing import List


@dataclass
class ARConfig:
    """UnifiedVoice (reference: tortoise/models/autoregressive.py:293-357, api.py:217-220)."""
    layers: int = 30
    model_dim: int = 1024
    heads: int = 16
    max_mel_tokens: int = 604
    max_text_tokens: int = 402
    max_conditioning_inputs: int = 2
    number_text_tokens: int = 255
    start_text_token: int = 255
    stop_text_token: int = 0
    number_mel_codes: int = 8194
    start_mel_token: int = 8192
    stop_mel_token: int = 8193
    mel_length_compression: int = 1024

    @property
    def mel_pos_len(self):  # autoregressive.py:340
        return self.max_mel_tokens + 2 + self.max_conditioning_inputs

    @property
    def text_pos_len(self):
        return self.max_text_tokens + 2

    @property
    def text_vocab(self):  # autoregressive.py:334 (types == 1)
        return self.number_text_tokens + 1


@dataclass
class DiffusionConfig:
    """DiffusionTts (reference: tortoise/models/diffusion_decoder.py:134-210, api.py:224-226)."""
    model_channels: int = 1024
    num_layers: int = 10
    in_channels: int = 100
    out_channels: int = 200
    in_latent_channels: int = 1024
    in_tokens: int = 8193
    num_heads: int = 16
    trained_steps: int = 4000  # api.py:64


@dataclass
class CLVPConfig:
    """CLVP with use_xformers=True (reference: tortoise/models/clvp.py:27-97, api.py:229-232)."""
    dim: int = 768
    dim_latent: int = 768
    depth: int = 20
    heads: int = 12
    num_text_tokens: int = 256
    num_speech_tokens: int = 8192
    ff_mult: int = 2
    rotary_dim: int = 32  # max(dim_head // 2, 32), xtransformers.py:781


@dataclass
class CVVPConfig:
    """CVVP as api.py builds it on demand (reference: tortoise/models/cvvp.py:63-98, api.py:252-257): two CollapsingTransformers
    (x-transformers Encoder with ff_mult = 1 -> 1 x 1 conv -> AttentionBlock -> 1 x 1 conv -> mean over time) and a latent projection each."""
    model_dim: int = 512
    heads: int = 8
    depth: int = 8           # conditioning_enc_depth == speech_enc_depth == 8
    mel_channels: int = 80
    mel_codes: int = 8192
    latent_multiplier: int = 1
    rotary_dim: int = 32     # max(dim_head // 2, 32), xtransformers.py:781

    @property
    def latent_dim(self):
        return self.latent_multiplier * self.model_dim


@dataclass
class VocoderConfig:
    """UnivNetGenerator (reference: tortoise/models/vocoder.py:225-265)."""
    noise_dim: int = 64
    channel_size: int = 32
    dilations: List[int] = field(default_factory=lambda: [1, 3, 9, 27])
    strides: List[int] = field(default_factory=lambda: [8, 8, 4])
    lrelu_slope: float = 0.2
    kpnet_conv_size: int = 3
    kpnet_hidden: int = 64
    hop_length: int = 256
    n_mel_channels: int = 100


@dataclass
class HifiganConfig:
    """HifiganGenerator of the streaming path (reference: tortoise/models/hifigan_decoder.py:159-294, api_fast.py:222-225)."""
    in_channels: int = 1024
    cond_channels: int = 1024
    upsample_initial_channel: int = 512
    upsample_factors: List[int] = field(default_factory=lambda: [8, 8, 2, 2])
    upsample_kernel_sizes: List[int] = field(default_factory=lambda: [16, 16, 4, 4])
    resblock_kernel_sizes: List[int] = field(default_factory=lambda: [3, 7, 11])
    resblock_dilation_sizes: List[int] = field(default_factory=lambda: [1, 3, 5])
    lrelu_slope: float = 0.1

    @property
    def hop(self):
        h = 1
        for u in self.upsample_factors:
            h *= u
        return h


PRESETS = {  # reference: tortoise/api.py:320-329
    "ultra_fast": {"num_autoregressive_samples": 16, "diffusion_iterations": 30, "cond_free": False},
    "fast": {"num_autoregressive_samples": 96, "diffusion_iterations": 80},
    "standard": {"num_autoregressive_samples": 256, "diffusion_iterations": 200},
    "high_quality": {"num_autoregressive_samples": 256, "diffusion_iterations": 400},
}
BASE_SETTINGS = {"temperature": 0.8, "length_penalty": 1.0, "repetition_penalty": 2.0,
                 "top_p": 0.8, "cond_free_k": 2.0, "diffusion_temperature": 1.0}

CALM_TOKEN = 83  # api.py:409
TACOTRON_MEL_MAX = 2.3143386840820312  # utils/audio.py:59-64
TACOTRON_MEL_MIN = -11.512925148010254
