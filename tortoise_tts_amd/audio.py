"""wav -> mel front-end of the voice_samples path (tortoise/api.py:258-299), in plain torch.

The reference builds the two conditioning mel spectrograms with torchaudio and librosa, neither of which exists in this
image, so their algorithms are restated here from the call sites and their documented definitions:

  * format_conditioning (tortoise/models/autoregressive.py... utils via api.py:22, 271-274): pad / crop the 22.05 kHz clip to
    132300 samples, TorchMelSpectrogram (tortoise/models/arch_util.py:295-331) = torchaudio MelSpectrogram(n_fft 1024, hop 256,
    power 2, 80 HTK-scale mels 0-8000 Hz with Slaney area normalisation) -> log(clamp(1e-5)) -> divide by data/mel_norms.pth;
  * diffusion conditioning (api.py:281-287): torchaudio.functional.resample 22050 -> 24000 (windowed-sinc, Hann, width 6,
    rolloff 0.99), pad / truncate to 102400 samples, TacotronSTFT(1024, 256, 1024, 100, 24000, 0, 12000)
    (tortoise/utils/audio.py:151-191) = |STFT| -> librosa Slaney-scale mel basis -> log(clamp(1e-5)).

PARITY: the third-party pieces (librosa==0.9.1 filters.mel, torchaudio melscale_fbanks / resample) are not vendored in the
reference tree; oracle/audio_oracle.py restates their published algorithms in loop-form numpy and is pinned against the
known-answer vectors of librosa's API documentation; tests/test_audio_frontend.py holds this module's filter banks and resampler
equal to that oracle, and runs the reference's own STFT / TacotronSTFT classes (librosa stubbed import-only) against
stft_magnitude + the clip + the log compression, so framing, window, padding and compression are pinned to the reference itself.
The encoders that consume the mels (csrc/cond.hip) are pinned against the reference modules; callers that have the reference's
own mels can pass them directly (TextToSpeech.get_conditioning_latents accepts (auto_mel, diffusion_mel) pairs).
This module is host-side glue (runs once per voice), not part of the accelerated path.
"""
import math
import os

import torch
import torch.nn.functional as F

AUTO_COND_SAMPLES = 132300     # api.py / autoregressive format_conditioning: ~6 s at 22.05 kHz
DIFF_COND_SAMPLES = 102400     # api.py:284


def _hz_to_mel(f, htk):
    f = torch.as_tensor(f, dtype=torch.float64)
    if htk:
        return 2595.0 * torch.log10(1.0 + f / 700.0)
    # Slaney: linear below 1 kHz (200/3 Hz per mel), logarithmic above
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0
    return torch.where(f >= min_log_hz, min_log_mel + torch.log(torch.clamp(f, min=1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m, htk):
    m = torch.as_tensor(m, dtype=torch.float64)
    if htk:
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0
    return torch.where(m >= min_log_mel, min_log_hz * torch.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax, htk):
    """Triangular mel filters [n_mels, n_fft // 2 + 1] with Slaney area normalisation (2 / bandwidth), on the HTK or the
    Slaney mel scale (librosa.filters.mel(norm='slaney') / torchaudio melscale_fbanks(norm='slaney'))."""
    freqs = torch.linspace(0, sr / 2, n_fft // 2 + 1, dtype=torch.float64)
    pts = _mel_to_hz(torch.linspace(float(_hz_to_mel(fmin, htk)), float(_hz_to_mel(fmax, htk)), n_mels + 2, dtype=torch.float64), htk)
    fdiff = pts[1:] - pts[:-1]
    ramps = pts[:, None] - freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    fb = torch.clamp(torch.minimum(lower, upper), min=0.0)
    fb = fb * (2.0 / (pts[2:] - pts[:-2]))[:, None]
    return fb.float()


def stft_magnitude(wav, n_fft=1024, hop=256, win=1024):
    """|STFT| [B, n_fft // 2 + 1, frames]: centred, reflect padding, periodic Hann window (tortoise/utils/stft.py, torchaudio Spectrogram)."""
    spec = torch.stft(wav, n_fft=n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win, periodic=True, device=wav.device),
                      center=True, pad_mode="reflect", return_complex=True)
    return spec.abs()


def pad_or_truncate(t, length):
    """tortoise/utils/audio.py pad_or_truncate: zero-pad or cut the last dimension to `length`."""
    if t.shape[-1] == length:
        return t
    if t.shape[-1] < length:
        return F.pad(t, (0, length - t.shape[-1]))
    return t[..., :length]


def resample_sinc(wav, orig_freq=22050, new_freq=24000, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional.resample (sinc_interp_hann): polyphase windowed-sinc kernel, one filter per output phase."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    if orig == new:
        return wav
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / orig)
    kern = kern.to(dtype=wav.dtype, device=wav.device)
    shape = wav.shape
    x = wav.reshape(-1, 1, shape[-1])
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x, kern, stride=orig)                      # [N, new, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(new * shape[-1] / orig))
    return y[..., :target].reshape(shape[:-1] + (target,))


def find_mel_norms(models_dir=None):
    """data/mel_norms.pth ships with the reference package (tortoise/data/); it is data, looked up, never copied."""
    cands = []
    if models_dir:
        cands.append(os.path.join(models_dir, "mel_norms.pth"))
    try:
        import tortoise
        cands.append(os.path.join(os.path.dirname(tortoise.__file__), "data", "mel_norms.pth"))
    except Exception:
        pass
    cands.append(os.path.join("/root/reference", "tortoise", "data", "mel_norms.pth"))
    for c in cands:
        if os.path.exists(c):
            return c
    return None


class MelFrontEnd:
    """Builds the (auto_mel [1, 80, T], diffusion_mel [1, 100, T]) pair of one 22.05 kHz clip (api.py:271-287)."""

    def __init__(self, models_dir=None, mel_norms=None):
        self.fb_auto = mel_filterbank(22050, 1024, 80, 0.0, 8000.0, htk=True)      # torchaudio MelSpectrogram default mel_scale
        self.fb_diff = mel_filterbank(24000, 1024, 100, 0.0, 12000.0, htk=False)   # librosa default
        if mel_norms is None:
            path = find_mel_norms(models_dir)
            if path is None:
                raise FileNotFoundError("mel_norms.pth (tortoise/data/, arch_util.py:290) was not found: pass models_dir= or mel_norms=, "
                                        "or give get_conditioning_latents ready (auto_mel, diffusion_mel) pairs")
            mel_norms = torch.load(path, map_location="cpu")
        self.mel_norms = torch.as_tensor(mel_norms).float()

    def auto_mel(self, clip, cond_length=AUTO_COND_SAMPLES, start=None):
        """format_conditioning: clips longer than cond_length are cropped at `start` (the reference draws it at random)."""
        clip = clip.float().reshape(1, -1)
        gap = clip.shape[-1] - cond_length
        if gap < 0:
            clip = F.pad(clip, (0, -gap))
        elif gap > 0:
            s = int(torch.randint(0, gap + 1, (1,))) if start is None else int(start)
            clip = clip[:, s:s + cond_length]
        power = stft_magnitude(clip) ** 2
        mel = torch.matmul(self.fb_auto.to(power.device), power)
        mel = torch.log(torch.clamp(mel, min=1e-5))
        return mel / self.mel_norms.to(mel.device)[None, :, None]

    def diffusion_mel(self, clip):
        clip = clip.float().reshape(1, -1)
        wav = pad_or_truncate(resample_sinc(clip, 22050, 24000), DIFF_COND_SAMPLES)
        mag = stft_magnitude(torch.clamp(wav, -1.0, 1.0))
        mel = torch.matmul(self.fb_diff.to(mag.device), mag)
        return torch.log(torch.clamp(mel, min=1e-5))

    def __call__(self, clip):
        return self.auto_mel(clip), self.diffusion_mel(clip)
