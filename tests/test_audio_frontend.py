"""CPU checks of the wav -> mel front-end restatement (tortoise_tts_amd/audio.py).  torchaudio / librosa are absent offline,
so the mel basis and the resampler are checked through DEFINITIONAL properties (mel-scale anchors, unit-area triangular filters,
a band-limited sine through the polyphase resampler, output shapes of api.py:271-287); the STFT, the clip and the log
compression around the basis ARE pinned against the reference's own TacotronSTFT / STFT classes (last test, reference tree only)."""
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


def _reference_audio_module():
    """tortoise/utils/audio.py + stft.py imported from the reference tree with import-only librosa stubs: pad_center / tiny /
    normalize are one-liners (the window already has the filter length), and librosa.filters.mel is handed OUR filter bank,
    so what the comparison pins is everything around the mel basis - the STFT (reflect padding, periodic Hann window,
    frame count), the clip to [-1, 1], the matmul and the log(clamp(1e-5)) compression - against the reference's own code."""
    import sys
    import types

    import numpy as np
    import pytest

    from oracle import ref_shims
    if not ref_shims.reference_available():
        pytest.skip("reference tree not present")
    ref_shims.install()
    if "librosa" not in sys.modules:
        lib, util, filt = types.ModuleType("librosa"), types.ModuleType("librosa.util"), types.ModuleType("librosa.filters")

        def pad_center(data, size=None, **kw):
            assert len(data) == size
            return data
        util.pad_center = pad_center
        util.tiny = lambda x: np.finfo(np.float32).tiny
        util.normalize = lambda x, norm=None: x
        filt.mel = lambda sr, n_fft, n_mels, fmin, fmax: A.mel_filterbank(sr, n_fft, n_mels, fmin, fmax, htk=False).numpy()
        lib.util, lib.filters = util, filt
        sys.modules.update({"librosa": lib, "librosa.util": util, "librosa.filters": filt})
    import importlib
    return importlib.import_module("tortoise.utils.audio")


def test_stft_and_compression_match_the_reference_tacotron_stft():
    RA = _reference_audio_module()
    g = torch.Generator().manual_seed(3)
    wav = (torch.randn(1, 30000, generator=g) * 0.4)                                 # some samples beyond [-1, 1]: exercises the clip
    for n_fft, hop in ((1024, 256), (800, 200)):
        ref_stft = RA.STFT(n_fft, hop, n_fft)
        mag_ref, _ = ref_stft.transform(wav)
        mag = A.stft_magnitude(wav, n_fft, hop, n_fft)
        assert mag.shape == mag_ref.shape
        assert (mag - mag_ref).abs().max() < 2e-3 * float(mag_ref.abs().max())       # conv1d against a float32 DFT basis vs torch.stft
    taco = RA.TacotronSTFT(1024, 256, 1024, 100, 24000, 0, 12000)                    # wav_to_univnet_mel's transform (audio.py:180-187)
    want = taco.mel_spectrogram(wav)
    fe = A.MelFrontEnd(mel_norms=torch.ones(80))
    mag = A.stft_magnitude(torch.clamp(wav, -1.0, 1.0))
    got = torch.log(torch.clamp(torch.matmul(fe.fb_diff, mag), min=1e-5))
    assert got.shape == want.shape == (1, 100, 30000 // 256 + 1)
    assert (got - want).abs().max() < 5e-3
    # normalize / denormalize_tacotron_mel (audio.py:55-64) are the affine pair the sampler's fused output uses
    m = torch.linspace(-11.0, 2.0, 50)
    assert torch.allclose(RA.denormalize_tacotron_mel(RA.normalize_tacotron_mel(m)), m, atol=1e-5)
    # pad_or_truncate (api.py:52-63) exec'd from the reference source
    import ast
    import os
    from oracle import ref_shims
    src = open(os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "api.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "pad_or_truncate")
    ns = {"torch": torch, "F": torch.nn.functional}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "api.py", "exec"), ns)
    for n in (10, 16, 20):
        t = torch.arange(float(n))[None]
        assert torch.equal(ns["pad_or_truncate"](t, 16), A.pad_or_truncate(t, 16))
