"""Window functions (mirror of python/mlx_audio/primitives/windows.py).

Taps are produced by the native builder `mia_generate_window` (float64 cosine
sums cast to float32, the arithmetic of windows.py:19-122) and cached on the
host; device copies live inside plans.
"""

from __future__ import annotations

import ctypes
from functools import lru_cache

import numpy as np
import torch

from .. import _lib

_SUPPORTED = ("bartlett", "blackman", "boxcar", "hamming", "hann", "hanning", "ones",
              "rectangular", "triangular")


@lru_cache(maxsize=128)
def _window_np_cached(name: str, n_fft: int, fftbins: bool) -> bytes:
    name = name.lower()
    if name not in _SUPPORTED:
        raise ValueError(
            f"Unknown window type: '{name}'. Supported: {', '.join(sorted(_SUPPORTED))}"
        )
    buf = (ctypes.c_float * max(n_fft, 1))()
    _lib.check(_lib.lib().mia_generate_window(name.encode(), n_fft, int(fftbins), buf))
    return np.frombuffer(buf, dtype=np.float32, count=n_fft).tobytes()


def window_np(window, n_fft: int, fftbins: bool = True) -> np.ndarray:
    """Host float32 taps for a window spec (str, ndarray or tensor)."""
    if isinstance(window, torch.Tensor):
        window = window.detach().cpu().numpy()
    if isinstance(window, np.ndarray):
        if window.shape[0] != n_fft:
            raise ValueError(
                f"Window array length ({window.shape[0]}) must match n_fft ({n_fft})"
            )
        return window.astype(np.float32)
    if not isinstance(window, str):
        raise TypeError(f"window must be str or mx.array, got {type(window).__name__}")
    return np.frombuffer(_window_np_cached(window, n_fft, bool(fftbins)), dtype=np.float32)


def padded_window_np(window, win_length: int, n_fft: int) -> np.ndarray:
    """Window of win_length centred in n_fft taps (reference stft.py:88-106)."""
    win = window_np(window, win_length, fftbins=True)
    if win_length < n_fft:
        left = (n_fft - win_length) // 2
        win = np.pad(win, (left, n_fft - win_length - left))
    return np.ascontiguousarray(win, dtype=np.float32)


def get_window(window, n_fft: int, fftbins: bool = True, device=None) -> torch.Tensor:
    """Get a window function as a float32 tensor of shape (n_fft,).

    Same names, aliases and errors as the reference `get_window`
    (windows.py:189-253).  `device` selects where the tensor lives (default:
    the current GPU if there is one, else the host).
    """
    w = torch.from_numpy(window_np(window, n_fft, fftbins).copy())
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    return w.to(device)
