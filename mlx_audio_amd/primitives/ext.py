"""Stand-alone ops with the names and argument order of the reference's native
module `mlx_audio.primitives._ext` (python/csrc/bindings.cpp:17-370), for
callers that used the extension directly."""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._runtime import as_device_f32, stream_ptr
from .mel import hz_to_mel, mel_filterbank, mel_to_hz  # noqa: F401
from .mfcc import dct as _dct
from .mfcc import dct_matrix_np
from .stft import _overlap_add_raw
from .windows import window_np


def overlap_add(frames, window, hop_length: int, output_length: int) -> torch.Tensor:
    """frames (batch, n_frames, n_fft), window (n_fft,) -> (batch, output_length)."""
    frames = as_device_f32(frames, ndim_allowed=(3,))
    window = as_device_f32(window, ndim_allowed=(1,)).to(frames.device)
    if window.shape[0] != frames.shape[2]:
        raise ValueError("window length must match frame length")
    if hop_length <= 0:
        raise ValueError(f"hop_length must be positive, got {hop_length}")
    return _overlap_add_raw(frames, window, int(hop_length), int(output_length))


def frame_signal(signal, frame_length: int, hop_length: int) -> torch.Tensor:
    """(batch, T) or (T,) -> (batch, n_frames, frame_length), materialised."""
    x = as_device_f32(signal)
    if x.dim() == 1:
        x = x[None]
    B, T = x.shape
    if frame_length <= 0:
        raise ValueError(f"frame_length must be positive, got {frame_length}")
    if hop_length <= 0:
        raise ValueError(f"hop_length must be positive, got {hop_length}")
    if T < frame_length:
        raise ValueError(
            f"Signal length ({T}) must be >= frame_length ({frame_length}). "
            "Consider padding the signal."
        )
    F = 1 + (T - frame_length) // hop_length
    out = torch.empty((B, F, frame_length), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mia_frame_signal(x.data_ptr(), B, T, int(frame_length),
                                               int(hop_length), out.data_ptr(),
                                               stream_ptr(x.device)))
    return out


def pad_signal(signal, pad_length: int, mode: str = "constant") -> torch.Tensor:
    """(batch, T) -> (batch, T + 2*pad_length); modes constant / edge / reflect."""
    x = as_device_f32(signal, ndim_allowed=(2,))
    if mode not in _lib.PAD_MODES:
        raise ValueError(f"Unknown pad_mode: '{mode}'. Supported: reflect, constant, edge")
    if pad_length == 0:
        return x
    B, T = x.shape
    out = torch.empty((B, T + 2 * pad_length), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mia_pad_signal(x.data_ptr(), B, T, int(pad_length),
                                             _lib.PAD_MODES[mode], out.data_ptr(),
                                             stream_ptr(x.device)))
    return out


def generate_window(window_type: str, length: int, periodic: bool = True) -> torch.Tensor:
    w = torch.from_numpy(window_np(window_type, length, periodic).copy())
    return w.to("cuda") if torch.cuda.is_available() else w


def get_dct_matrix(n_out: int, n_in: int, norm: str = "ortho") -> torch.Tensor:
    m = torch.from_numpy(dct_matrix_np(n_out, n_in, norm if norm else None).copy())
    return m.to("cuda") if torch.cuda.is_available() else m


def dct(x, n: int = -1, axis: int = -1, norm: str = "ortho") -> torch.Tensor:
    return _dct(x, type=2, n=None if n < 0 else n, axis=axis, norm=norm if norm else None)
