"""CPU oracle for Griffin-Lim (TEST INFRASTRUCTURE, not product code).

NumPy restatement of `mlx_audio.primitives.griffinlim` (reference
python/mlx_audio/primitives/griffinlim.py): griffinlim (:17-196) and griffinlim_iter
(:199-284), on top of oracle.spectral's stft / istft.  PARITY UNPINNED (MLX arithmetic, no
stored numbers); the random initial phase is NumPy's in the reference too
(`np.random.default_rng(random_state).uniform`, :112-115), so a seeded run starts identically.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""

from __future__ import annotations

import numpy as np

from . import spectral as sp


def _fit_frames(C, n_frames):
    if C.shape[-1] > n_frames:
        return C[..., :n_frames]
    if C.shape[-1] < n_frames:
        return np.pad(C, [(0, 0)] * (C.ndim - 1) + [(0, n_frames - C.shape[-1])])
    return C


def griffinlim(S, n_iter=32, hop_length=None, win_length=None, n_fft=None, window="hann",
               center=True, length=None, pad_mode="constant", momentum=0.99, init="random",
               random_state=None):
    if n_iter <= 0:
        raise ValueError(f"n_iter must be positive, got {n_iter}")
    if momentum < 0.0:
        raise ValueError(f"momentum must be >= 0.0, got {momentum}")
    if momentum >= 1.0:
        raise ValueError(f"momentum must be < 1.0, got {momentum}")
    S = np.asarray(S, np.float32)
    batched = S.ndim == 3
    if not batched:
        S = S[None]
    B, bins, F = S.shape
    n_fft = 2 * (bins - 1) if n_fft is None else n_fft
    hop_length = n_fft // 4 if hop_length is None else hop_length
    win_length = n_fft if win_length is None else win_length
    rng = np.random.default_rng(random_state)
    if init == "random":
        angles = rng.uniform(-np.pi, np.pi, (B, bins, F)).astype(np.float32)
    elif init == "zeros":
        angles = np.zeros((B, bins, F), np.float32)
    else:
        raise ValueError(f"Unknown init: '{init}'. Supported: 'random', 'zeros'")
    rebuilt = (S * np.exp(1j * angles.astype(np.complex64))).astype(np.complex64)
    tprev = rebuilt
    kw = dict(hop_length=hop_length, win_length=win_length, n_fft=n_fft, window=window, center=center)
    for _ in range(n_iter):
        y = sp.istft(rebuilt, length=length, **kw)
        new = _fit_frames(sp.stft(y, pad_mode=pad_mode, **kw), F)
        ang = np.arctan2(new.imag, new.real).astype(np.float32)
        new = (S * np.exp(1j * ang.astype(np.complex64))).astype(np.complex64)
        if momentum > 0:
            rebuilt = (new + np.float32(momentum) * (new - tprev)).astype(np.complex64)
            tprev = new
        else:
            rebuilt = new
    y = sp.istft(rebuilt, length=length, **kw)
    return y if batched else y[0]


def griffinlim_iter(S, angles, hop_length, win_length, n_fft, window="hann", center=True,
                    pad_mode="constant", momentum=0.99, tprev=None):
    S = np.asarray(S, np.float32)
    angles = np.asarray(angles, np.float32)
    rebuilt = (S * np.exp(1j * angles.astype(np.complex64))).astype(np.complex64)
    kw = dict(hop_length=hop_length, win_length=win_length, n_fft=n_fft, window=window, center=center)
    y = sp.istft(rebuilt, **kw)
    new = sp.stft(y, pad_mode=pad_mode, **kw)
    err = np.mean((S - np.abs(new)) ** 2, dtype=np.float32)
    ang = np.arctan2(new.imag, new.real).astype(np.float32)
    new = (S * np.exp(1j * ang.astype(np.complex64))).astype(np.complex64)
    if momentum > 0 and tprev is not None:
        out = (new + np.float32(momentum) * (new - np.asarray(tprev, np.complex64))).astype(np.complex64)
    else:
        out = new
    return ang, out, err
