"""The key / shape manifests of tortoise_tts_amd/weights.py against the LIVE reference modules: a synthetic state_dict built from a
manifest must load into the reference class with strict=True at the api.py:217-236 / api_fast.py:222-225 hyper-parameters (i.e. the
engine consumes exactly the checkpoints' keys).  Build container only (the reference tree is absent on the GPU box)."""
import pytest
import torch

from oracle import ref_shims
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.config import ARConfig, CLVPConfig, CVVPConfig, DiffusionConfig, HifiganConfig, VocoderConfig

pytestmark = pytest.mark.skipif(not ref_shims.reference_available(), reason="reference tree not present")


def _meta_load(build, manifest):
    with torch.device("meta"):
        m = build()
    want = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    have = {k: tuple(s) for k, s in manifest.items()}
    missing = sorted(set(want) - set(have))
    extra = sorted(set(have) - set(want))
    assert not missing and not extra, (missing[:8], extra[:8])
    bad = {k: (have[k], want[k]) for k in want if have[k] != want[k]}
    assert not bad, list(bad.items())[:8]


def test_manifests_match_the_reference_modules():
    ref = ref_shims.import_reference()
    a = ARConfig()
    _meta_load(lambda: ref.UnifiedVoice(max_mel_tokens=a.max_mel_tokens, max_text_tokens=a.max_text_tokens,
                                        max_conditioning_inputs=a.max_conditioning_inputs, layers=a.layers, model_dim=a.model_dim,
                                        heads=a.heads, number_text_tokens=a.number_text_tokens, start_text_token=a.start_text_token,
                                        checkpointing=False, train_solo_embeddings=False), W.ar_manifest(a))
    d = DiffusionConfig()
    _meta_load(lambda: ref.DiffusionTts(model_channels=d.model_channels, num_layers=d.num_layers, in_channels=d.in_channels,
                                        out_channels=d.out_channels, in_latent_channels=d.in_latent_channels, in_tokens=d.in_tokens,
                                        dropout=0, use_fp16=False, num_heads=d.num_heads, layer_drop=0, unconditioned_percentage=0),
               W.diffusion_manifest(d))
    c = CLVPConfig()
    _meta_load(lambda: ref.CLVP(dim_text=c.dim, dim_speech=c.dim, dim_latent=c.dim_latent, num_text_tokens=c.num_text_tokens,
                                text_enc_depth=c.depth, text_seq_len=350, text_heads=c.heads, num_speech_tokens=c.num_speech_tokens,
                                speech_enc_depth=c.depth, speech_heads=c.heads, speech_seq_len=430, use_xformers=True),
               W.clvp_manifest(c))
    from tortoise.models.cvvp import CVVP
    v = CVVPConfig()
    _meta_load(lambda: CVVP(model_dim=v.model_dim, transformer_heads=v.heads, dropout=0, mel_codes=v.mel_codes, conditioning_enc_depth=v.depth,
                            cond_mask_percentage=0, speech_enc_depth=v.depth, speech_mask_percentage=0, latent_multiplier=v.latent_multiplier),
               W.cvvp_manifest(v))  # api.py:254-255
    _meta_load(lambda: ref.UnivNetGenerator(), W.vocoder_manifest(VocoderConfig()))
    from tortoise.models.hifigan_decoder import HifiganGenerator
    from tortoise.models.random_latent_generator import RandomLatentConverter
    h = HifiganConfig()
    _meta_load(lambda: HifiganGenerator(in_channels=h.in_channels, out_channels=1, resblock_type="1",
                                        resblock_dilation_sizes=[list(h.resblock_dilation_sizes)] * 3,
                                        resblock_kernel_sizes=list(h.resblock_kernel_sizes), upsample_kernel_sizes=list(h.upsample_kernel_sizes),
                                        upsample_initial_channel=h.upsample_initial_channel, upsample_factors=list(h.upsample_factors),
                                        cond_channels=h.cond_channels), W.hifigan_manifest(h))
    for ch in (1024, 2048):
        _meta_load(lambda: RandomLatentConverter(ch), W.rlg_manifest(ch))
