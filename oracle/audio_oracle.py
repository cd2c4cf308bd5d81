"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, loop form) of the third-party pieces of the reference's wav -> mel
conditioning front-end.  Nothing in the product path imports this file.

The reference builds its two conditioning spectrograms through dependencies that are NOT vendored in /root/reference and are
not installable offline:

  * `librosa.filters.mel` — librosa==0.9.1 (reference requirements.txt:10), called by TacotronSTFT
    (tortoise/utils/audio.py:151-178: `librosa_mel_fn(sr=sampling_rate, n_fft=filter_length, n_mels=n_mel_channels,
    fmin=mel_fmin, fmax=mel_fmax)`, i.e. the Slaney mel scale (htk=False) with norm='slaney');
  * `torchaudio.transforms.MelSpectrogram(..., norm='slaney')` with the default `mel_scale='htk'` — torchaudio (unpinned in
    requirements.txt), called by TorchMelSpectrogram (tortoise/models/arch_util.py:295-331).  torchaudio's
    `functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm, mel_scale)` is the same triangular
    construction as librosa's (torchaudio's own test-suite asserts `melscale_fbanks(...) == librosa.filters.mel(..., htk=
    (mel_scale == 'htk'), norm=norm).T`), so one restatement with an `htk` switch covers both call sites;
  * `torchaudio.functional.resample(wav, 22050, 24000)` (tortoise/api.py:281-283): polyphase windowed-sinc interpolation,
    `sinc_interp_hann`, lowpass_filter_width=6, rolloff=0.99.

What follows restates their PUBLISHED algorithms.  The mel functions are pinned (tests/test_audio_frontend.py) against the
known-answer vectors printed in librosa's API documentation for hz_to_mel / mel_to_hz / mel_frequencies / filters.mel; the
resampler has no published vector, so it is anchored on its defining properties and on agreement between this loop form and
the product's vectorised form (two independent derivations of the same published kernel formula).
"""
import math

import numpy as np


# ------------------------------------------------------------------------------------------------ mel scales
def hz_to_mel(f, htk=False):
    """librosa.core.convert.hz_to_mel (0.9.1): HTK: 2595 log10(1 + f / 700).  Slaney (Auditory Toolbox): linear below 1 kHz at
    200/3 Hz per mel, above it 27 log-spaced mels per factor of 6.4."""
    f = np.asarray(f, dtype=np.float64)
    if htk:
        return 2595.0 * np.log10(1.0 + f / 700.0)
    out = f / (200.0 / 3.0)
    hi = f >= 1000.0
    if np.any(hi):
        out = np.where(hi, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) / (math.log(6.4) / 27.0), out)
    return out


def mel_to_hz(m, htk=False):
    m = np.asarray(m, dtype=np.float64)
    if htk:
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    out = (200.0 / 3.0) * m
    hi = m >= 15.0
    if np.any(hi):
        out = np.where(hi, 1000.0 * np.exp((math.log(6.4) / 27.0) * (m - 15.0)), out)
    return out


def mel_frequencies(n_mels=128, fmin=0.0, fmax=11025.0, htk=False):
    """n_mels frequencies uniformly spaced on the mel axis between fmin and fmax (librosa.mel_frequencies)."""
    lo, hi = float(hz_to_mel(fmin, htk)), float(hz_to_mel(fmax, htk))
    return mel_to_hz(np.linspace(lo, hi, n_mels), htk)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk, norm='slaney') -> float32 [n_mels, 1 + n_fft // 2]: filter i is the
    triangle over (f[i], f[i+1], f[i+2]) of the n_mels + 2 mel-spaced frequencies, scaled by 2 / (f[i+2] - f[i]) (unit area in Hz).
    Written filter by filter and bin by bin (the product builds it with broadcast ramps)."""
    if fmax is None:
        fmax = sr / 2.0
    nbin = 1 + n_fft // 2
    fft_f = [k * (sr / 2.0) / (nbin - 1) for k in range(nbin)]
    pts = mel_frequencies(n_mels + 2, fmin, fmax, htk)
    w = np.zeros((n_mels, nbin), dtype=np.float64)
    for i in range(n_mels):
        left, centre, right = pts[i], pts[i + 1], pts[i + 2]
        for k, f in enumerate(fft_f):
            up = (f - left) / (centre - left)
            down = (right - f) / (right - centre)
            w[i, k] = max(0.0, min(up, down)) * 2.0 / (right - left)
    return w.astype(np.float32)


# ------------------------------------------------------------------------------------------------ resampler
def resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional._get_sinc_resample_kernel (sinc_interp_hann) in loop form -> (kernel [new][taps] float64, width).
    With orig, new reduced by their gcd, output sample j of phase p = j mod new is sum_i kernel[p][i] * x[(j // new) * orig + i - width]:
    a sinc at the cutoff base = min(orig, new) * rolloff, under a Hann window lowpass_filter_width zero crossings wide, scaled by
    base / orig."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    taps = 2 * width + orig
    k = np.zeros((new, taps), dtype=np.float64)
    for p in range(new):
        for i in range(taps):
            t = (-p / new + (i - width) / orig) * base
            t = max(-lowpass_filter_width, min(lowpass_filter_width, t))
            win = math.cos(t * math.pi / lowpass_filter_width / 2.0) ** 2
            x = t * math.pi
            k[p, i] = (1.0 if x == 0.0 else math.sin(x) / x) * win * base / orig
    return k, width, orig, new


def resample(wav, orig_freq=22050, new_freq=24000, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional.resample on a 1-D float array: zero-pad (width, width + orig), stride-orig correlation with the
    `new` phase filters, interleave the phases, keep ceil(new * n / orig) samples."""
    wav = np.asarray(wav, dtype=np.float64)
    k, width, orig, new = resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff)
    n = wav.shape[-1]
    x = np.concatenate([np.zeros(width), wav, np.zeros(width + orig)])
    frames = (x.shape[0] - k.shape[1]) // orig + 1
    out = np.zeros(frames * new, dtype=np.float64)
    for fr in range(frames):
        seg = x[fr * orig: fr * orig + k.shape[1]]
        out[fr * new:(fr + 1) * new] = k @ seg
    return out[: int(math.ceil(new * n / orig))]
