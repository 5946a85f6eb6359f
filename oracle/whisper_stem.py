"""CPU oracle for the Whisper encoder stem (TEST INFRASTRUCTURE, not product code).

A NumPy restatement of what `AudioEncoder.__call__` does before its transformer blocks
(reference `python/mlx_audio/models/whisper/layers/encoder.py:72-109`, positional embedding
`:16-35`).  The arithmetic of the reference lives in MLX (`mlx.nn.Conv1d`, `mlx.nn.gelu`),
which is absent here; what those two do is public and restated below:

  * `mlx.nn.Conv1d(in, out, kernel_size, stride, padding)` takes channels-last input
    (batch, length, in), holds its weight as (out, kernel_size, in) and computes
    y[b, t, o] = bias[o] + sum_{k, i} w[o, k, i] * x_padded[b, stride * t + k, i]
    with `padding` zero rows on both ends of the length axis (cross-correlation, no flip);
  * `mlx.nn.gelu(x)` = x * (1 + erf(x / sqrt(2))) / 2, the exact form.

PARITY UNPINNED: the reference holds no stored numbers for this stage.  The restatement is
pinned by `torch.nn.functional.conv1d` + `torch.nn.functional.gelu` (CPU, float64) in
tests/test_whisper_stem_host.py - an independent implementation of the same published
definitions.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""

from __future__ import annotations

import math

import numpy as np
from scipy.special import erf


def sinusoids(length: int, dim: int, max_timescale: float = 10000.0) -> np.ndarray:
    """encoder.py:16-35: [sin(pos * inv_timescale) | cos(pos * inv_timescale)], float32."""
    half = dim // 2
    log_timescale = math.log(max_timescale) / (half - 1)
    inv = np.exp(-log_timescale * np.arange(half, dtype=np.float32)).astype(np.float32)
    scaled = np.arange(length, dtype=np.float32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(scaled), np.cos(scaled)], axis=1).astype(np.float32)


def gelu(x: np.ndarray) -> np.ndarray:
    """mlx.nn.gelu: x * (1 + erf(x / sqrt(2))) / 2."""
    return (x * (1.0 + erf(x / math.sqrt(2.0))) / 2.0).astype(x.dtype)


def conv1d_channels_last(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int, padding: int) -> np.ndarray:
    """x (B, L, Cin), w (Cout, K, Cin), b (Cout,) -> (B, (L + 2 padding - K) // stride + 1, Cout)."""
    B, L, _ = x.shape
    cout, K, cin = w.shape
    xp = np.pad(x, ((0, 0), (padding, padding), (0, 0)))
    t_out = (L + 2 * padding - K) // stride + 1
    out = np.zeros((B, t_out, cout), dtype=x.dtype)
    for k in range(K):  # one matrix product per tap
        rows = xp[:, k:k + stride * (t_out - 1) + 1:stride, :]
        out += rows @ w[:, k, :].T
    return out + b[None, None, :]


def encoder_stem(mel: np.ndarray, conv1_w, conv1_b, conv2_w, conv2_b, positional_embedding=None,
                 dtype=np.float32) -> np.ndarray:
    """encoder.py:87-109: mel (B, n_mels, T) or (n_mels, T) -> (B, T', n_state)."""
    mel = np.asarray(mel, dtype=dtype)
    if mel.ndim == 2:
        mel = mel[None]
    x = mel.transpose(0, 2, 1)  # encoder.py:97-99
    x = gelu(conv1d_channels_last(x, np.asarray(conv1_w, dtype), np.asarray(conv1_b, dtype), 1, 1))
    x = gelu(conv1d_channels_last(x, np.asarray(conv2_w, dtype), np.asarray(conv2_b, dtype), 2, 1))
    if positional_embedding is not None:
        x = x + np.asarray(positional_embedding, dtype)[None, :x.shape[1], :]  # encoder.py:107-109
    return x
