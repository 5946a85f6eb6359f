"""CPU oracle for the spectral features (TEST INFRASTRUCTURE, not product code).

NumPy restatement of `mlx_audio.primitives.features` (reference
python/mlx_audio/primitives/features.py): spectral_centroid (:59-136), spectral_bandwidth
(:139-241), spectral_rolloff (:244-382), spectral_flatness (:385-464) and
zero_crossing_rate (:642-769).  float32 arithmetic like the reference's arrays.

PARITY UNPINNED: the reference computes these with MLX reductions (absent here) and its
tests compare against librosa (absent here) at run time, holding no stored numbers.  The
restatement is pinned by closed-form cases in tests/test_features_host.py only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""

from __future__ import annotations

import numpy as np

from . import spectral as sp

DIVISION_EPSILON = 1e-8  # constants/algorithms.py:44


def get_frequencies(sr, n_fft):
    """features.py:21-23."""
    return np.linspace(0, sr / 2.0, n_fft // 2 + 1).astype(np.float32)


def compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode, power=1.0):
    """features.py:26-56."""
    if S is not None:
        return np.asarray(S, np.float32)
    if y is None:
        raise ValueError("Either y (audio) or S (spectrogram) must be provided")
    Sc = sp.stft(np.asarray(y, np.float32), n_fft=n_fft, hop_length=hop_length,
                 win_length=win_length, window=window, center=center, pad_mode=pad_mode)
    S = np.abs(Sc).astype(np.float32)
    if power != 1.0:
        S = np.power(S, np.float32(power)).astype(np.float32)
    return S


def _batched(S):
    return (S, True) if S.ndim == 3 else (S[None], False)


def spectral_centroid(y=None, sr=22050, S=None, n_fft=2048, hop_length=512, win_length=None,
                      window="hann", center=True, pad_mode="constant", freq=None):
    S = compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode)
    freq = get_frequencies(sr, n_fft) if freq is None else np.asarray(freq, np.float32)
    S, batched = _batched(S)
    num = np.sum(freq[None, :, None] * S, axis=1, keepdims=True, dtype=np.float32)
    den = np.sum(S, axis=1, keepdims=True, dtype=np.float32) + np.float32(DIVISION_EPSILON)
    out = (num / den).astype(np.float32)
    return out if batched else out[0]


def spectral_bandwidth(y=None, sr=22050, S=None, n_fft=2048, hop_length=512, win_length=None,
                       window="hann", center=True, pad_mode="constant", freq=None, centroid=None,
                       p=2.0, norm=True):
    S = compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode)
    freq = get_frequencies(sr, n_fft) if freq is None else np.asarray(freq, np.float32)
    S, batched = _batched(S)
    if centroid is None:
        centroid = spectral_centroid(S=S, sr=sr, n_fft=n_fft, freq=freq)
    centroid = np.asarray(centroid, np.float32)
    if centroid.ndim == 2:
        centroid = centroid[None]
    dev = np.abs(freq[None, :, None] - centroid)
    w = np.sum(S * np.power(dev, np.float32(p)), axis=1, keepdims=True, dtype=np.float32)
    if norm:
        w = w / (np.sum(S, axis=1, keepdims=True, dtype=np.float32) + np.float32(DIVISION_EPSILON))
    out = np.power(w, np.float32(1.0 / p)).astype(np.float32)
    return out if batched else out[0]


def spectral_rolloff(y=None, sr=22050, S=None, n_fft=2048, hop_length=512, win_length=None,
                     window="hann", center=True, pad_mode="constant", freq=None, roll_percent=0.85):
    if roll_percent < 0.0:
        raise ValueError(f"roll_percent must be >= 0.0, got {roll_percent}")
    if roll_percent > 1.0:
        raise ValueError(f"roll_percent must be <= 1.0, got {roll_percent}")
    S = compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode)
    freq = get_frequencies(sr, n_fft) if freq is None else np.asarray(freq, np.float32)
    S, batched = _batched(S)
    cs = np.cumsum(S, axis=1, dtype=np.float32)
    thr = np.float32(roll_percent) * cs[:, -1:, :]
    exceeds = cs >= thr
    idx = np.argmax(exceeds, axis=1)
    idx[~exceeds.any(axis=1)] = S.shape[1] - 1
    out = freq[idx][:, None, :].astype(np.float32)
    return out if batched else out[0]


def spectral_flatness(y=None, S=None, n_fft=2048, hop_length=512, win_length=None, window="hann",
                      center=True, pad_mode="constant", power=2.0, amin=1e-10):
    S = compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode, power)
    S, batched = _batched(S)
    S = np.maximum(S, np.float32(amin))
    g = np.exp(np.mean(np.log(S), axis=1, keepdims=True, dtype=np.float32))
    a = np.mean(S, axis=1, keepdims=True, dtype=np.float32)
    out = (g / (a + np.float32(DIVISION_EPSILON))).astype(np.float32)
    return out if batched else out[0]


def zero_crossing_rate(y, frame_length=2048, hop_length=512, center=True, pad_mode="edge",
                       use_mlx=True):
    if frame_length <= 0:
        raise ValueError(f"frame_length must be positive, got {frame_length}")
    if hop_length <= 0:
        raise ValueError(f"hop_length must be positive, got {hop_length}")
    y = np.asarray(y, np.float32)
    is_1d = y.ndim == 1
    if is_1d:
        y = y[None]
    if center:
        if pad_mode not in ("constant", "edge"):
            raise ValueError(f"Unknown pad_mode: '{pad_mode}'. Supported: 'constant', 'edge'")
        y = np.pad(y, [(0, 0), (frame_length // 2, frame_length // 2)], mode=pad_mode)
    n_frames = 1 + (y.shape[1] - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n_frames)[:, None]
    frames = y[:, idx]
    sign = (frames >= 0) if use_mlx else ~np.signbit(frames)
    cross = sign[..., 1:] != sign[..., :-1]
    zcr = (cross.sum(axis=-1).astype(np.float32) / np.float32(frame_length))[:, None, :]
    return zcr[0] if is_1d else zcr


def contrast_bands(freq, fmin, n_bands, quantile):
    """Band table of spectral_contrast (features.py:566-599): per octave band the boolean mask
    of its bins (after the neighbour-bin and last-band extensions), whether the last masked
    bin is dropped, and the number of values averaged at each end."""
    freq = np.asarray(freq)
    octa = np.zeros(n_bands + 2)
    octa[1:] = fmin * (2.0 ** np.arange(0, n_bands + 1))
    bands = []
    for k, (f_low, f_high) in enumerate(zip(octa[:-1], octa[1:])):
        mask = np.logical_and(freq >= f_low, freq <= f_high)
        idx = np.flatnonzero(mask)
        if len(idx) == 0:
            bands.append(None)
            continue
        if k > 0 and idx[0] > 0:
            mask[idx[0] - 1] = True
        if k == n_bands and idx[-1] + 1 < len(mask):
            mask[idx[-1] + 1:] = True
        n_quantile = int(np.maximum(np.rint(quantile * np.sum(mask)), 1))
        sel = np.flatnonzero(mask)
        if k < n_bands and len(sel) > 1:
            sel = sel[:-1]
        bands.append((sel, n_quantile))
    return bands


def spectral_contrast(y=None, sr=22050, S=None, n_fft=2048, hop_length=512, win_length=None,
                      window="hann", center=True, pad_mode="constant", freq=None, fmin=200.0,
                      n_bands=6, quantile=0.02, linear=False):
    """features.py:467-639 (the NumPy branch, which the reference describes as librosa's exact
    algorithm)."""
    if n_bands <= 0:
        raise ValueError(f"n_bands must be positive, got {n_bands}")
    if quantile < 0.0:
        raise ValueError(f"quantile must be >= 0.0, got {quantile}")
    if quantile > 1.0:
        raise ValueError(f"quantile must be <= 1.0, got {quantile}")
    S = compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode)
    freq = get_frequencies(sr, n_fft) if freq is None else np.asarray(freq, np.float32)
    S, batched = _batched(S)
    B, _, F = S.shape
    valley = np.zeros((B, n_bands + 1, F), np.float32)
    peak = np.zeros_like(valley)
    for k, band in enumerate(contrast_bands(freq, fmin, n_bands, quantile)):
        if band is None:
            continue
        sel, nq = band
        sub = S[:, sel, :]
        n_sub = sub.shape[1]
        if n_sub == 0:
            continue
        if nq < n_sub:
            bottom = np.partition(sub, nq - 1, axis=1)
            valley[:, k, :] = np.mean(bottom[:, :nq, :], axis=1)
            top = np.partition(sub, n_sub - nq, axis=1)
            peak[:, k, :] = np.mean(top[:, n_sub - nq:, :], axis=1)
        else:
            valley[:, k, :] = np.mean(sub, axis=1)
            peak[:, k, :] = np.mean(sub, axis=1)
    if linear:
        out = peak - valley
    else:
        out = 10.0 * np.log10(np.maximum(peak, 1e-10) / 1.0) - 10.0 * np.log10(np.maximum(valley, 1e-10) / 1.0)
    out = out.astype(np.float32)
    return out if batched else out[0]
