"""CPU oracle for onset strength / detection (TEST INFRASTRUCTURE, not product code).

NumPy restatement of `mlx_audio.primitives.onset` (reference
python/mlx_audio/primitives/onset.py): onset_strength (:17-178), onset_detect (:181-322),
onset_strength_multi (:325-420).  The local-max and running-mean filters are SciPy's
(`scipy.ndimage.maximum_filter1d / uniform_filter1d`, called by the reference at :154-166);
the rest is float32 NumPy.  PARITY UNPINNED for the MLX part (mel spectrogram, log1p, mean /
max over the mel axis): no stored numbers in the reference's tests.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""

from __future__ import annotations

import numpy as np

from . import spectral as sp


def onset_strength(y=None, sr=22050, S=None, n_fft=2048, hop_length=512, win_length=None,
                   window="hann", center=True, pad_mode="constant", lag=1, max_size=1,
                   detrend=False, aggregate="mean", n_mels=128, fmin=0.0, fmax=None):
    if lag <= 0:
        raise ValueError(f"lag must be positive, got {lag}")
    if max_size <= 0:
        raise ValueError(f"max_size must be positive, got {max_size}")
    if S is None:
        if y is None:
            raise ValueError("Either y (audio) or S (spectrogram) must be provided")
        S = sp.melspectrogram(np.asarray(y, np.float32), sr=sr, n_fft=n_fft, hop_length=hop_length,
                              win_length=win_length, window=window, center=center,
                              pad_mode=pad_mode, n_mels=n_mels, fmin=fmin, fmax=fmax, power=1.0)
    S = np.asarray(S, np.float32)
    is_1d = S.ndim == 2
    if is_1d:
        S = S[None]
    S_db = np.log1p(S * np.float32(1e4)).astype(np.float32)
    d = np.maximum(S_db[..., lag:] - S_db[..., :-lag], np.float32(0.0))
    if aggregate == "mean":
        env = np.mean(d, axis=1, dtype=np.float32)
    elif aggregate == "median":
        env = np.median(d, axis=1).astype(np.float32)
    elif aggregate == "max":
        env = np.max(d, axis=1)
    else:
        raise ValueError(f"Unknown aggregate: '{aggregate}'. Use 'mean', 'median', or 'max'")
    if max_size > 1:
        from scipy.ndimage import maximum_filter1d

        env = maximum_filter1d(env, size=max_size, axis=-1).astype(np.float32)
    if detrend:
        from scipy.ndimage import uniform_filter1d

        k = max(1, env.shape[-1] // 10)
        env = np.maximum((env - uniform_filter1d(env, size=k, axis=-1)).astype(np.float32),
                         np.float32(0.0))
    env = np.pad(env, [(0, 0), (lag, 0)], mode="constant")
    return env[0] if is_1d else env


def peak_pick(env, pre_max=1, post_max=1, pre_avg=1, post_avg=1, delta=0.07, wait=1):
    """onset.py:271-299."""
    env = np.asarray(env)
    n = len(env)
    onsets = []
    last = -wait - 1
    for i in range(n):
        if env[i] < np.max(env[max(0, i - pre_max):min(n, i + post_max + 1)]):
            continue
        if env[i] < np.mean(env[max(0, i - pre_avg):min(n, i + post_avg + 1)]) + delta:
            continue
        if i < last + wait:
            continue
        onsets.append(i)
        last = i
    return np.array(onsets, dtype=np.int64)


def onset_detect(y=None, sr=22050, onset_envelope=None, hop_length=512, backtrack=False,
                 units="frames", pre_max=1, post_max=1, pre_avg=1, post_avg=1, delta=0.07, wait=1,
                 **kwargs):
    if onset_envelope is None:
        if y is None:
            raise ValueError("Either y (audio) or onset_envelope must be provided")
        onset_envelope = onset_strength(y, sr=sr, hop_length=hop_length, **kwargs)
    env = np.asarray(onset_envelope)
    if env.ndim > 1:
        env = env[0]
    onsets = peak_pick(env, pre_max, post_max, pre_avg, post_avg, delta, wait)
    if backtrack and len(onsets) > 0:
        for i, o in enumerate(onsets):
            start = max(0, o - pre_max * 2)
            if start < o:
                onsets[i] = start + np.argmin(env[start:o])
    if units == "frames":
        return onsets
    if units == "samples":
        return onsets * hop_length
    if units == "time":
        return (onsets * hop_length) / sr
    raise ValueError(f"Unknown units: '{units}'. Use 'frames', 'samples', or 'time'")
