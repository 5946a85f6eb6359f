"""Mel filterbank and mel spectrogram (mirror of python/mlx_audio/primitives/mel.py).

`melspectrogram` is ONE fused kernel: pad + frame + window + real FFT +
|X|^power + band-sparse filterbank projection; the complex spectrum never
reaches HBM.
"""

from __future__ import annotations

import ctypes
from functools import lru_cache

import numpy as np
import torch

from .. import _lib
from .._runtime import as_device_f32, get_plan, stream_ptr
from .stft import _resolve
from .windows import padded_window_np

DEFAULT_N_MELS = 128


def _convert(fn_name: str, values, htk: bool) -> np.ndarray:
    arr = np.ascontiguousarray(np.asarray(values, dtype=np.float64))
    out = np.empty_like(arr)
    dptr = ctypes.POINTER(ctypes.c_double)
    _lib.check(getattr(_lib.lib(), fn_name)(arr.ctypes.data_as(dptr), arr.size, int(bool(htk)),
                                            out.ctypes.data_as(dptr)))
    return out


def hz_to_mel(frequencies, htk: bool = False) -> np.ndarray:
    """Hz -> mel, Slaney (default) or HTK; NumPy in and out (reference mel.py:42-73)."""
    return _convert("mia_hz_to_mel", frequencies, htk)


def mel_to_hz(mels, htk: bool = False) -> np.ndarray:
    """mel -> Hz (reference mel.py:76-104)."""
    return _convert("mia_mel_to_hz", mels, htk)


@lru_cache(maxsize=64)
def _mel_filterbank_bytes(sr, n_fft, n_mels, fmin, fmax, htk, norm) -> bytes:
    if norm == "slaney":
        norm_code = 1
    elif norm is None:
        norm_code = 0
    else:
        raise ValueError(f"Unknown norm: '{norm}'. Supported: 'slaney', None")
    bins = 1 + n_fft // 2
    out = np.empty((n_mels, bins), dtype=np.float32)
    _lib.check(_lib.lib().mia_mel_filterbank(
        int(sr), int(n_fft), int(n_mels), float(fmin), float(fmax), int(bool(htk)), norm_code,
        out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
    return out.tobytes()


def mel_filterbank_np(sr, n_fft, n_mels=DEFAULT_N_MELS, fmin=0.0, fmax=None, htk=False,
                      norm="slaney") -> np.ndarray:
    """Host float32 filterbank (n_mels, n_fft//2+1); validation of mel.py:227-237."""
    if n_mels <= 0:
        raise ValueError(f"n_mels must be positive, got {n_mels}")
    if fmin < 0:
        raise ValueError(f"fmin must be non-negative, got {fmin}")
    if fmax is None:
        fmax = sr / 2.0
    if fmin >= fmax:
        raise ValueError(f"fmin ({fmin}) must be less than fmax ({fmax})")
    if fmax > sr / 2.0:
        raise ValueError(f"fmax ({fmax}) cannot exceed Nyquist frequency ({sr / 2.0})")
    data = _mel_filterbank_bytes(sr, n_fft, n_mels, float(fmin), float(fmax), bool(htk), norm)
    return np.frombuffer(data, dtype=np.float32).reshape(n_mels, 1 + n_fft // 2)


def mel_filterbank(sr: int, n_fft: int, n_mels: int = DEFAULT_N_MELS, fmin: float = 0.0,
                   fmax=None, htk: bool = False, norm="slaney", device=None) -> torch.Tensor:
    """Mel filterbank matrix (n_mels, n_fft//2+1), float32 (reference mel.py:182-253)."""
    fb = torch.from_numpy(mel_filterbank_np(sr, n_fft, n_mels, fmin, fmax, htk, norm).copy())
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    return fb.to(device)


def _mel_raw(y2d, sr, n_fft, hop_length, win_length, window, center, pad_mode, power, n_mels,
             fmin, fmax, htk, norm, epilogue=_lib.EPI_NONE, ref=1.0, amin=1e-10, coef=10.0,
             top_db=-1.0, per_clip=False, frames_first=False) -> torch.Tensor:
    """(B, T) float32 on GPU -> (B, n_mels, frames) float32 via the fused kernel.  With
    `frames_first` the result is (B, frames, n_mels): written that way by the kernel where the
    plan can (contiguous), else a transposed view of the (B, n_mels, frames) result."""
    hop_length, win_length = _resolve(n_fft, hop_length, win_length)
    if center and pad_mode not in _lib.PAD_MODES:
        raise ValueError(f"Unknown pad_mode: '{pad_mode}'. Supported: reflect, constant, edge")
    basis = mel_filterbank_np(sr, n_fft, n_mels, fmin, fmax, htk, norm)
    win = padded_window_np(window, win_length, n_fft)
    plan = get_plan(y2d.device, n_fft, hop_length, center, pad_mode if center else "constant",
                    win, mel_basis=basis, power=power)
    B, T = y2d.shape
    F = plan.num_frames(T)
    lib = _lib.lib()
    direct = bool(frames_first) and lib.mia_mel_frames_supported(plan.handle) == 1
    out = torch.empty((B, F, n_mels) if direct else (B, n_mels, F), dtype=torch.float32, device=y2d.device)
    ws_bytes = lib.mia_mel_workspace_bytes(plan.handle, B, T)
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=y2d.device)
    entry = lib.mia_melspectrogram_frames if direct else lib.mia_melspectrogram
    with torch.cuda.device(y2d.device):
        _lib.check(entry(
            plan.handle, y2d.data_ptr(), B, T, int(epilogue), float(ref), float(amin),
            float(coef), float(top_db), int(bool(per_clip)), out.data_ptr(), ws.data_ptr(),
            ws_bytes, stream_ptr(y2d.device)))
    return out.transpose(1, 2) if (frames_first and not direct) else out


def melspectrogram(y, sr: int = 22050, n_fft: int = 2048, hop_length=None, win_length=None,
                   window="hann", center: bool = True, pad_mode: str = "constant",
                   power: float = 2.0, n_mels: int = DEFAULT_N_MELS, fmin: float = 0.0,
                   fmax=None, htk: bool = False, norm="slaney") -> torch.Tensor:
    """Mel spectrogram (reference mel.py:281-373).

    y: (samples,) or (batch, samples) -> (n_mels, n_frames) or
    (batch, n_mels, n_frames), frames contiguous.
    """
    y = as_device_f32(y)
    is_1d = y.dim() == 1
    if is_1d:
        y = y[None, :]
    out = _mel_raw(y, sr, n_fft, hop_length, win_length, window, center, pad_mode, power,
                   n_mels, fmin, fmax, htk, norm)
    return out[0] if is_1d else out
