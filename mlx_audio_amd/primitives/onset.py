"""Onset strength and detection on MI355X with the signatures of
`mlx_audio.primitives.onset` (reference python/mlx_audio/primitives/onset.py).

`onset_strength` runs on the GPU (mel spectrogram, log compression, lagged positive
difference, aggregation over the bands, optional local-max and detrending filters);
`onset_detect` picks peaks on the host, as the reference does (a sequential scan over a
few hundred envelope values).
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._runtime import require_gpu, stream_ptr
from .mel import melspectrogram

_AGG = {"mean": 0, "max": 1, "median": 2}


def onset_strength(y=None, sr: int = 22050, S=None, n_fft: int = 2048, hop_length: int = 512,
                   win_length=None, window="hann", center: bool = True,
                   pad_mode: str = "constant", lag: int = 1, max_size: int = 1,
                   detrend: bool = False, aggregate: str = "mean", n_mels: int = 128,
                   fmin: float = 0.0, fmax=None) -> torch.Tensor:
    """Onset strength envelope from spectral flux (reference onset.py:17-178)."""
    if lag <= 0:
        raise ValueError(f"lag must be positive, got {lag}")
    if max_size <= 0:
        raise ValueError(f"max_size must be positive, got {max_size}")
    if S is None:
        if y is None:
            raise ValueError("Either y (audio) or S (spectrogram) must be provided")
        S = melspectrogram(y, sr=sr, n_fft=n_fft, hop_length=hop_length, win_length=win_length,
                           window=window, center=center, pad_mode=pad_mode, n_mels=n_mels,
                           fmin=fmin, fmax=fmax, power=1.0)
    else:
        require_gpu()
        if not isinstance(S, torch.Tensor):
            S = torch.as_tensor(np.asarray(S))
        if not S.is_cuda:
            S = S.to("cuda")
        S = S.to(torch.float32)
    if aggregate not in _AGG:
        raise ValueError(f"Unknown aggregate: '{aggregate}'. Use 'mean', 'median', or 'max'")
    is_1d = S.dim() == 2
    Sb = (S[None] if is_1d else S).contiguous()
    rows, bands, frames = Sb.shape
    dev = Sb.device
    lib = _lib.lib()
    if frames < lag:
        # Both slices of the lagged difference are empty (onset.py:131), every later step keeps
        # the empty envelope, and the closing pad puts `lag` zeros in front of it
        # (onset.py:170-171): the result has `lag` frames, not `frames`.
        env = torch.zeros((rows, int(lag)), dtype=torch.float32, device=dev)
        return env[0] if is_1d else env
    env = torch.empty((rows, frames), dtype=torch.float32, device=dev)
    n = frames - lag   # the envelope proper; the first `lag` frames are the zero padding
    with torch.cuda.device(dev):
        stream = stream_ptr(dev)
        _lib.check(lib.mia_onset_flux(Sb.data_ptr(), rows, bands, frames, int(lag),
                                      _AGG[aggregate], env.data_ptr(), stream))
        steps = []
        if max_size > 1:
            steps.append((int(max_size), 0))
        if detrend:
            steps.append((max(1, n // 10), 1))
        for size, mode in steps:
            if n == 0:
                break
            nxt = torch.zeros_like(env)
            _lib.check(lib.mia_envelope_filter(env.data_ptr() + 4 * lag, rows, n, frames, size,
                                               mode, nxt.data_ptr() + 4 * lag, frames, stream))
            env = nxt
    return env[0] if is_1d else env


def onset_detect(y=None, sr: int = 22050, onset_envelope=None, hop_length: int = 512,
                 backtrack: bool = False, units: str = "frames", pre_max: int = 1,
                 post_max: int = 1, pre_avg: int = 1, post_avg: int = 1, delta: float = 0.07,
                 wait: int = 1, **kwargs) -> torch.Tensor:
    """Peak picking on the onset envelope (reference onset.py:181-322).  As in the reference
    the scan runs on the host and, for a batched envelope, looks at the first row only; the
    result has the types `mx.array` gives the reference's NumPy result: int64 frame / sample
    indices as they are, times narrowed from float64 to float32."""
    if hop_length <= 0:
        raise ValueError(f"hop_length must be positive, got {hop_length}")
    if onset_envelope is None:
        if y is None:
            raise ValueError("Either y (audio) or onset_envelope must be provided")
        onset_envelope = onset_strength(y, sr=sr, hop_length=hop_length, **kwargs)
    if isinstance(onset_envelope, torch.Tensor):
        env = onset_envelope.detach().cpu().numpy()
    else:
        env = np.asarray(onset_envelope)
    if env.ndim > 1:
        env = env[0]
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
    onsets = np.array(onsets, dtype=np.int64)
    if backtrack and len(onsets) > 0:
        for k, o in enumerate(onsets):
            start = max(0, o - pre_max * 2)
            if start < o:
                onsets[k] = start + np.argmin(env[start:o])
    if units == "frames":
        result = onsets
    elif units == "samples":
        result = onsets * hop_length
    elif units == "time":
        result = (onsets * hop_length) / sr
    else:
        raise ValueError(f"Unknown units: '{units}'. Use 'frames', 'samples', or 'time'")
    result = np.asarray(result)
    return torch.from_numpy(result.astype(np.float32) if result.dtype.kind == "f" else result)


def onset_strength_multi(y, sr: int = 22050, n_fft: int = 2048, hop_length: int = 512,
                         channels=None, aggregate: str = "mean", **kwargs) -> torch.Tensor:
    """Onset strength per frequency channel (reference onset.py:325-420).  The reference
    computes one 128-band mel spectrogram and hands it to `onset_strength` together with each
    channel's (fmin, fmax); a spectrogram passed in is used as it is, so every channel gets the
    same envelope.  That behaviour is kept."""
    if channels is None:
        nyquist = sr / 2
        channels = [(20.0, 200.0), (200.0, 2000.0), (2000.0, nyquist)]
    S = melspectrogram(y, sr=sr, n_fft=n_fft, hop_length=hop_length, n_mels=128, power=1.0, **kwargs)
    is_1d = S.dim() == 2
    if is_1d:
        S = S[None]
    envs = [onset_strength(S=S, sr=sr, hop_length=hop_length, aggregate=aggregate, fmin=lo, fmax=hi,
                           n_mels=128) for lo, hi in channels]
    out = torch.stack(envs, dim=-2)
    return out[0] if is_1d else out
