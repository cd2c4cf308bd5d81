"""CPU checks of the wav -> mel front-end restatement (tortoise_tts_amd/audio.py).  torchaudio / librosa are absent offline,
so these are DEFINITIONAL properties (the module header says "parity unpinned"): mel-scale anchors, unit-area triangular
filters, the polyphase resampler on a band-limited signal, output shapes of api.py:271-287."""
import math

import torch

from tortoise_tts_amd import audio as A


def test_mel_scales_hit_their_anchor_points():
    assert abs(float(A._hz_to_mel(1000.0, htk=False)) - 15.0) < 1e-9                 # Slaney: 1 kHz = 15 mel (200/3 Hz per mel)
    assert abs(float(A._hz_to_mel(6400.0, htk=False)) - 42.0) < 1e-9                 # 27 log-spaced steps per factor 6.4
    assert abs(float(A._hz_to_mel(1000.0, htk=True)) - 2595.0 * math.log10(1.0 + 1000.0 / 700.0)) < 1e-9
    for htk in (False, True):
        f = torch.tensor([0.0, 123.0, 999.0, 1000.0, 4321.0, 11999.0], dtype=torch.float64)
        assert torch.allclose(A._mel_to_hz(A._hz_to_mel(f, htk), htk), f, atol=1e-6)


def test_filterbanks_are_slaney_normalised_triangles():
    for sr, n_mels, fmax, htk in ((22050, 80, 8000.0, True), (24000, 100, 12000.0, False)):
        fb = A.mel_filterbank(sr, 1024, n_mels, 0.0, fmax, htk)
        assert fb.shape == (n_mels, 513) and (fb >= 0).all()
        peaks = fb.argmax(dim=1)
        assert (peaks[1:] >= peaks[:-1]).all()                                       # centres ascend
        # unit area in Hz (Slaney normalisation): sum over bins * bin width ~= 1 for filters several bins wide
        area = fb.double().sum(dim=1) * (sr / 1024)
        wide = (fb > 0).sum(dim=1) >= 6
        assert wide.any() and torch.allclose(area[wide], torch.ones_like(area[wide]), atol=0.08)
        above = fb[:, int(fmax / (sr / 2) * 512) + 2:]
        assert above.numel() == 0 or above.abs().max() == 0                          # nothing above fmax


def test_resampler_keeps_a_band_limited_sine():
    sr0, sr1, f0 = 22050, 24000, 440.0
    t0 = torch.arange(sr0, dtype=torch.float64) / sr0
    y = A.resample_sinc(torch.sin(2 * math.pi * f0 * t0).float()[None], sr0, sr1)
    assert y.shape[-1] == math.ceil(sr0 * 160 / 147) == sr1
    t1 = torch.arange(sr1, dtype=torch.float64) / sr1
    want = torch.sin(2 * math.pi * f0 * t1).float()
    assert (y[0, 200:-200] - want[200:-200]).abs().max() < 2e-3                      # interior (edges see the zero padding)


def test_front_end_shapes_match_the_reference_call_sites():
    fe = A.MelFrontEnd(mel_norms=torch.ones(80))
    g = torch.Generator().manual_seed(0)
    clip = torch.randn(1, 150000, generator=g).clamp(-1, 1) * 0.3
    am, dm = fe.auto_mel(clip, start=0), fe.diffusion_mel(clip)
    assert am.shape == (1, 80, A.AUTO_COND_SAMPLES // 256 + 1)                       # 517 frames (centred STFT)
    assert dm.shape == (1, 100, A.DIFF_COND_SAMPLES // 256 + 1)                      # 401 frames
    assert torch.isfinite(am).all() and torch.isfinite(dm).all()
    assert float(dm.min()) >= math.log(1e-5) - 1e-6
    short = fe.auto_mel(clip[:, :1000])                                              # zero-padded to the conditioning length
    assert short.shape == am.shape
