"""STFT / iSTFT (mirror of python/mlx_audio/primitives/stft.py).

`stft` is one fused kernel (pad + frame + window + real FFT); `istft` is an
inverse real FFT kernel followed by the deterministic gather overlap-add.
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._runtime import as_device_f32, get_plan, require_gpu, stream_ptr
from .windows import padded_window_np, window_np


def _resolve(n_fft, hop_length, win_length):
    if hop_length is None:
        hop_length = n_fft // 4
    if win_length is None:
        win_length = n_fft
    if hop_length <= 0:
        raise ValueError(f"hop_length must be positive, got {hop_length}")
    if win_length <= 0:
        raise ValueError(f"win_length must be positive, got {win_length}")
    if win_length > n_fft:
        raise ValueError(f"win_length ({win_length}) must be <= n_fft ({n_fft})")
    if hop_length > n_fft:
        raise ValueError(
            f"hop_length ({hop_length}) should typically be <= n_fft ({n_fft})"
        )
    return hop_length, win_length


def _stft_raw(y2d: torch.Tensor, n_fft, hop_length, win_length, window, center, pad_mode,
              scale: float = 1.0) -> torch.Tensor:
    """(B, T) float32 on GPU -> (B, frames, bins) complex64 (physical layout)."""
    win = padded_window_np(window, win_length, n_fft)
    plan = get_plan(y2d.device, n_fft, hop_length, center, pad_mode if center else "constant", win)
    B, T = y2d.shape
    F = plan.num_frames(T)
    out = torch.empty((B, F, n_fft // 2 + 1), dtype=torch.complex64, device=y2d.device)
    lib = _lib.lib()
    # 0 bytes for the compiled FFT sizes; other lengths run on the any-length engine, which
    # materialises its frames in a workspace
    ws_bytes = lib.mia_stft_workspace_bytes(plan.handle, B, T)
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=y2d.device)
    with torch.cuda.device(y2d.device):
        _lib.check(lib.mia_stft_ws(plan.handle, y2d.data_ptr(), B, T, float(scale),
                                   out.data_ptr(), ws.data_ptr(), ws_bytes,
                                   stream_ptr(y2d.device)))
    return out


def stft(y, n_fft: int = 2048, hop_length=None, win_length=None, window="hann",
         center: bool = True, pad_mode: str = "constant") -> torch.Tensor:
    """Short-Time Fourier Transform (reference stft.py:136-222).

    y: (samples,) or (batch, samples).  Returns complex64 of shape
    (n_fft//2+1, n_frames) or (batch, n_fft//2+1, n_frames) — as in the
    reference, a transposed view of a (batch, n_frames, bins) buffer.
    """
    hop_length, win_length = _resolve(n_fft, hop_length, win_length)
    if center and pad_mode not in _lib.PAD_MODES:
        raise ValueError(f"Unknown pad_mode: '{pad_mode}'. Supported: reflect, constant, edge")
    y = as_device_f32(y)
    is_1d = y.dim() == 1
    if is_1d:
        y = y[None, :]
    out = _stft_raw(y, n_fft, hop_length, win_length, window, center, pad_mode)
    out = out.transpose(1, 2)
    return out[0] if is_1d else out


def _overlap_add_raw(frames, window_dev, hop_length, output_length, scale=1.0):
    B, F, n_fft = frames.shape
    out = torch.empty((B, output_length), dtype=torch.float32, device=frames.device)
    with torch.cuda.device(frames.device):
        _lib.check(_lib.lib().mia_overlap_add(frames.data_ptr(), window_dev.data_ptr(), B, F,
                                              n_fft, hop_length, output_length, float(scale),
                                              out.data_ptr(), stream_ptr(frames.device)))
    return out


def _istft_raw(spec: torch.Tensor, n_fft, hop_length, win_length, window, padded_length,
               scale: float = 1.0) -> torch.Tensor:
    """(B, bins, F) complex64 with arbitrary strides -> (B, padded_length) float32."""
    win = padded_window_np(window, win_length, n_fft)
    plan = get_plan(spec.device, n_fft, hop_length, True, "constant", win)
    B, bins, F = spec.shape
    if bins != n_fft // 2 + 1:
        raise ValueError(f"stft_matrix has {bins} bins but n_fft={n_fft} needs {n_fft // 2 + 1}")
    lib = _lib.lib()
    ws_bytes = lib.mia_istft_workspace_bytes(plan.handle, B, F)
    if (ws_bytes == 0 and F > 1 and bins > 1 and spec.stride(2) == 1 and spec.stride(1) == F
            and spec.stride(0) == bins * F):
        # frames contiguous (the reference's layout) and a plan whose inverse is the fused kernel
        # (no workspace: the overlap stays in LDS), which reads whole rows of bins and is several
        # times slower through strides: one transposing pass first
        rows_first = torch.empty((B, F, bins), dtype=torch.complex64, device=spec.device)
        with torch.cuda.device(spec.device):
            _lib.check(lib.mia_transpose_spectrum(spec.data_ptr(), B, bins, F, rows_first.data_ptr(),
                                                  stream_ptr(spec.device)))
        spec = rows_first.transpose(1, 2)
    out = torch.empty((B, padded_length), dtype=torch.float32, device=spec.device)
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=spec.device)
    sb, sk, sf = spec.stride()
    with torch.cuda.device(spec.device):
        _lib.check(lib.mia_istft(plan.handle, spec.data_ptr(), B, F, sb, sk, sf, padded_length,
                                 float(scale), out.data_ptr(), ws.data_ptr(), ws_bytes,
                                 stream_ptr(spec.device)))
    return out


def istft(stft_matrix, hop_length=None, win_length=None, n_fft=None, window="hann",
          center: bool = True, length=None) -> torch.Tensor:
    """Inverse STFT (reference stft.py:225-344)."""
    require_gpu()
    if not isinstance(stft_matrix, torch.Tensor):
        stft_matrix = torch.as_tensor(np.asarray(stft_matrix))
    if stft_matrix.dim() not in (2, 3):
        raise ValueError(f"stft_matrix must be 2D or 3D, got {stft_matrix.dim()}D")
    if not stft_matrix.is_cuda:
        stft_matrix = stft_matrix.to("cuda")
    if stft_matrix.dtype != torch.complex64:
        stft_matrix = stft_matrix.to(torch.complex64)
    is_2d = stft_matrix.dim() == 2
    if is_2d:
        stft_matrix = stft_matrix[None]
    _, freq_bins, n_frames = stft_matrix.shape
    if n_fft is None:
        n_fft = 2 * (freq_bins - 1)
    if hop_length is None:
        hop_length = n_fft // 4
    if win_length is None:
        win_length = n_fft
    if length is not None:
        padded_length = length + n_fft if center else length
    else:
        padded_length = n_fft + (n_frames - 1) * hop_length
    y = _istft_raw(stft_matrix, n_fft, hop_length, win_length, window, padded_length)
    if center:
        pad = n_fft // 2
        if length is not None:
            y = y[:, pad:pad + length]
        else:
            end = y.shape[1] - pad
            y = y[:, pad:end] if end > pad else y[:, :0]
    elif length is not None:
        cur = y.shape[1]
        if length < cur:
            y = y[:, :length]
        elif length > cur:
            y = torch.nn.functional.pad(y, (0, length - cur))
    return y[0] if is_2d else y


def _complex_elementwise(fn_name: str, stft_matrix: torch.Tensor) -> torch.Tensor:
    require_gpu()
    z = stft_matrix if stft_matrix.is_cuda else stft_matrix.to("cuda")
    z = z.to(torch.complex64)
    # What `stft` returns is a (.., bins, frames) view of a (.., frames, bins) buffer: transpose
    # while taking the magnitude / phase instead of copying the complex spectrum first.
    if z.dim() in (2, 3) and z.numel() > 0 and not z.is_contiguous():
        zt = z.transpose(-1, -2)
        rows = 1 if z.dim() == 2 else z.shape[0]
        if zt.is_contiguous() and rows <= 65535:
            frames, bins = zt.shape[-2], zt.shape[-1]
            out = torch.empty(z.shape, dtype=torch.float32, device=z.device)
            with torch.cuda.device(z.device):
                _lib.check(_lib.lib().mia_spectrum_part(
                    zt.data_ptr(), rows, frames, bins, 0 if fn_name == "mia_magnitude" else 1,
                    out.data_ptr(), stream_ptr(z.device)))
            return out
    z = z.contiguous()
    out = torch.empty(z.shape, dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(getattr(_lib.lib(), fn_name)(z.data_ptr(), z.numel(), out.data_ptr(),
                                                stream_ptr(z.device)))
    return out


def magnitude(stft_matrix: torch.Tensor) -> torch.Tensor:
    """|S| (reference stft.py:347-362)."""
    return _complex_elementwise("mia_magnitude", stft_matrix)


def phase(stft_matrix: torch.Tensor) -> torch.Tensor:
    """atan2(imag, real) in radians (reference stft.py:365-379)."""
    return _complex_elementwise("mia_phase", stft_matrix)


def check_nola(window, hop_length: int, n_fft: int, tol: float = 1e-10) -> bool:
    """Nonzero-overlap-add check, host side (reference stft.py:382-431)."""
    win = window_np(window, n_fft, fftbins=True)
    step = hop_length
    n_bins = n_fft // step
    binsums = sum(win[i * step:(i + 1) * step] ** 2 for i in range(n_bins))
    if n_fft % step != 0:
        binsums[: n_fft % step] += win[-(n_fft % step):] ** 2
    return bool(np.min(binsums) > tol)
