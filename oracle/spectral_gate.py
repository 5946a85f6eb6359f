"""CPU oracle for spectral gating (TEST INFRASTRUCTURE, not product code).

NumPy restatement of `mlx_audio.primitives.spectral_gate` (reference
python/mlx_audio/primitives/spectral_gate.py): spectral_gate (:19-238) and
spectral_gate_adaptive (:241-353).  The reference does this arithmetic in NumPy and
`scipy.ndimage` (minimum_filter1d, uniform_filter) on the host between its STFT and inverse
STFT; the same calls are made here, around oracle.spectral's stft / istft.  PARITY UNPINNED for
the STFT pair (MLX); the gating in between is the reference's own host arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""

from __future__ import annotations

import numpy as np
from scipy.ndimage import minimum_filter1d, uniform_filter

from . import spectral as sp

DIVISION_EPSILON = 1e-8


def _check(n_fft, hop_length, prop_decrease):
    if n_fft <= 0:
        raise ValueError(f"n_fft must be positive, got {n_fft}")
    if hop_length <= 0:
        raise ValueError(f"hop_length must be positive, got {hop_length}")
    if prop_decrease < 0.0:
        raise ValueError(f"prop_decrease must be >= 0.0, got {prop_decrease}")
    if prop_decrease > 1.0:
        raise ValueError(f"prop_decrease must be <= 1.0, got {prop_decrease}")


def gate_mask(audio_mag, noise_mag=None, *, sr=22050, n_fft=2048, hop_length=512,
              threshold_db=-20.0, prop_decrease=1.0, n_std_thresh=1.5, time_constant_s=0.05,
              freq_smooth_hz=100.0, time_smooth_ms=10.0, stationary=True, n_grad_freq=2,
              n_grad_time=4):
    """spectral_gate.py:137-215: the mask from the magnitudes (B, bins, frames)."""
    audio_mag = np.asarray(audio_mag, np.float32)
    B = audio_mag.shape[0]
    if noise_mag is not None:
        noise_mag = np.asarray(noise_mag, np.float32)
        thr = np.mean(noise_mag, axis=-1, keepdims=True) + n_std_thresh * np.std(noise_mag, axis=-1, keepdims=True)
    elif stationary:
        energy = np.mean(audio_mag ** 2, axis=1)
        F = energy.shape[-1]
        n_quiet = max(1, int(0.1 * F))
        rows = []
        for b in range(B):
            quiet = audio_mag[b][:, np.argsort(energy[b])[:n_quiet]]
            rows.append(np.mean(quiet, axis=1, keepdims=True) + n_std_thresh * np.std(quiet, axis=1, keepdims=True))
        thr = np.stack(rows, axis=0)
    else:
        tc = max(1, int(time_constant_s * sr / hop_length))
        thr = minimum_filter1d(audio_mag, size=tc, axis=-1, mode="nearest") * (1 + n_std_thresh)
    thr = thr * 10 ** (threshold_db / 20.0)
    mask = np.clip((audio_mag - thr) / (audio_mag + DIVISION_EPSILON), 0.0, 1.0)
    if freq_smooth_hz > 0 or time_smooth_ms > 0:
        fb = max(1, int(freq_smooth_hz * n_fft / sr))
        tf = max(1, int(time_smooth_ms * sr / (1000 * hop_length)))
        mask = uniform_filter(mask, size=(1, fb, tf), mode="nearest")
    if n_grad_freq > 1 or n_grad_time > 1:
        mask = uniform_filter(mask, size=(1, n_grad_freq, n_grad_time), mode="nearest")
    return 1.0 - prop_decrease * (1.0 - mask)


def spectral_gate(audio, noise_profile=None, *, sr=22050, n_fft=2048, hop_length=None,
                  win_length=None, window="hann", **kw):
    hop_length = n_fft // 4 if hop_length is None else hop_length
    win_length = n_fft if win_length is None else win_length
    _check(n_fft, hop_length, kw.get("prop_decrease", 1.0))
    audio = np.asarray(audio, np.float32)
    is_1d = audio.ndim == 1
    if is_1d:
        audio = audio[None]
    n = audio.shape[-1]
    st = dict(n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window, center=True)
    C = sp.stft(audio, **st)
    mag, ph = np.abs(C).astype(np.float32), np.arctan2(C.imag, C.real).astype(np.float32)
    noise_mag = None
    if noise_profile is not None:
        noise = np.asarray(noise_profile, np.float32)
        noise_mag = np.abs(sp.stft(noise if noise.ndim > 1 else noise[None], **st)).astype(np.float32)
    mask = gate_mask(mag, noise_mag, sr=sr, n_fft=n_fft, hop_length=hop_length, **kw)
    out = sp.istft((mag * mask * np.exp(1j * ph)).astype(np.complex64), hop_length=hop_length,
                   win_length=win_length, window=window, center=True, length=n)
    return out[0] if is_1d else out


def adaptive_mask(audio_mag, *, sr=22050, hop_length=512, threshold_db=-20.0, prop_decrease=1.0,
                  look_ahead_ms=100.0, look_back_ms=100.0):
    """spectral_gate.py:303-337."""
    audio_mag = np.asarray(audio_mag, np.float32)
    ahead = max(1, int(look_ahead_ms * sr / (1000 * hop_length)))
    back = max(1, int(look_back_ms * sr / (1000 * hop_length)))
    floor = minimum_filter1d(audio_mag, size=ahead + back + 1, axis=-1, mode="nearest")
    thr = floor * (1.0 + 10 ** (threshold_db / 20.0))
    mask = np.clip((audio_mag - thr) / (audio_mag + DIVISION_EPSILON), 0.0, 1.0)
    mask = 1.0 - prop_decrease * (1.0 - mask)
    return uniform_filter(mask, size=(1, 3, 5), mode="nearest")


def spectral_gate_adaptive(audio, *, sr=22050, n_fft=2048, hop_length=None, threshold_db=-20.0,
                           prop_decrease=1.0, look_ahead_ms=100.0, look_back_ms=100.0):
    hop_length = n_fft // 4 if hop_length is None else hop_length
    _check(n_fft, hop_length, prop_decrease)
    audio = np.asarray(audio, np.float32)
    is_1d = audio.ndim == 1
    if is_1d:
        audio = audio[None]
    n = audio.shape[-1]
    C = sp.stft(audio, n_fft=n_fft, hop_length=hop_length, center=True)
    mag, ph = np.abs(C).astype(np.float32), np.arctan2(C.imag, C.real).astype(np.float32)
    mask = adaptive_mask(mag, sr=sr, hop_length=hop_length, threshold_db=threshold_db,
                         prop_decrease=prop_decrease, look_ahead_ms=look_ahead_ms,
                         look_back_ms=look_back_ms)
    out = sp.istft((mag * mask * np.exp(1j * ph)).astype(np.complex64), hop_length=hop_length,
                   center=True, length=n)
    return out[0] if is_1d else out
