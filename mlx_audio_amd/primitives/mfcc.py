"""DCT-II and MFCC (mirror of python/mlx_audio/primitives/mfcc.py)."""

from __future__ import annotations

import ctypes
from functools import lru_cache

import numpy as np
import torch

from .. import _lib
from .._runtime import as_device_f32, get_plan, require_gpu, stream_ptr
from .mel import mel_filterbank_np
from .stft import _resolve
from .windows import padded_window_np


@lru_cache(maxsize=32)
def _dct_matrix_bytes(n_out: int, n_in: int, norm) -> bytes:
    out = np.empty((n_out, n_in), dtype=np.float32)
    _lib.check(_lib.lib().mia_dct_matrix(n_out, n_in, int(norm == "ortho"),
                                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
    return out.tobytes()


def dct_matrix_np(n_out: int, n_in: int, norm="ortho") -> np.ndarray:
    """DCT-II basis (n_out, n_in): float64 cosines cast to float32 (mfcc.py:23-66)."""
    return np.frombuffer(_dct_matrix_bytes(n_out, n_in, norm), dtype=np.float32).reshape(
        n_out, n_in)


_dct_dev_cache: dict = {}


def _dct_matrix_dev(n_out, n_in, norm, device) -> torch.Tensor:
    key = (n_out, n_in, norm, device.index)
    t = _dct_dev_cache.get(key)
    if t is None:
        t = torch.from_numpy(dct_matrix_np(n_out, n_in, norm).copy()).to(device)
        _dct_dev_cache[key] = t
    return t


def dct(x, type: int = 2, n=None, axis: int = -1, norm="ortho") -> torch.Tensor:
    """DCT-II along `axis` with a cached basis (reference mfcc.py:69-140)."""
    if type != 2:
        raise ValueError(f"Only DCT type 2 is supported, got {type}")
    require_gpu()
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x))
    if not x.is_cuda:
        x = x.to("cuda")
    x = x.to(torch.float32).contiguous()
    axis = axis % x.dim()
    n_in = x.shape[axis]
    if n is None:
        n = n_in
    outer = int(np.prod(x.shape[:axis], dtype=np.int64))
    inner = int(np.prod(x.shape[axis + 1:], dtype=np.int64))
    basis = _dct_matrix_dev(int(n), int(n_in), norm, x.device)
    out_shape = list(x.shape)
    out_shape[axis] = int(n)
    out = torch.empty(out_shape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mia_dct2(x.data_ptr(), outer, n_in, inner, basis.data_ptr(),
                                       int(n), out.data_ptr(), stream_ptr(x.device)))
    return out


def _lifter_np(n_mfcc: int, lifter: int):
    if lifter <= 0:
        return None
    k = np.arange(n_mfcc)
    return (1 + (lifter / 2.0) * np.sin(np.pi * (k + 1) / lifter)).astype(np.float32)


def mfcc(y=None, sr: int = 22050, S=None, n_mfcc: int = 20, dct_type: int = 2, norm="ortho",
         lifter: int = 0, n_fft: int = 2048, hop_length: int = 512, win_length=None,
         window="hann", center: bool = True, pad_mode: str = "constant", power: float = 2.0,
         n_mels: int = 128, fmin: float = 0.0, fmax=None, htk: bool = False,
         mel_norm="slaney", per_clip: bool = False) -> torch.Tensor:
    """Mel-frequency cepstral coefficients (reference mfcc.py:143-281).

    From a waveform this is two launches: the fused mel/dB kernel and a DCT
    kernel that applies the top_db floor while staging its input.  The floor
    uses the maximum of the whole batch (reference behaviour) unless
    `per_clip=True`.
    """
    if n_mfcc <= 0:
        raise ValueError(f"n_mfcc must be positive, got {n_mfcc}")
    if dct_type != 2:
        raise ValueError(f"Only DCT type 2 is supported, got {dct_type}")
    lift = _lifter_np(n_mfcc, lifter)
    if S is not None:  # already log-power: DCT (+ lifter) only
        if not isinstance(S, torch.Tensor):
            S = torch.as_tensor(np.asarray(S))
        batched = S.dim() == 3
        Sb = S if batched else S[None]
        M = dct(Sb, type=dct_type, n=n_mfcc, axis=1, norm=norm)
        if lift is not None:
            M = M * torch.from_numpy(lift).to(M.device)[:, None]
        return M if batched else M[0]
    y = as_device_f32(y)
    is_1d = y.dim() == 1
    if is_1d:
        y = y[None, :]
    hop_length, win_length = _resolve(n_fft, hop_length, win_length)
    if center and pad_mode not in _lib.PAD_MODES:
        raise ValueError(f"Unknown pad_mode: '{pad_mode}'. Supported: reflect, constant, edge")
    basis = mel_filterbank_np(sr, n_fft, n_mels, fmin, fmax, htk, mel_norm)
    win = padded_window_np(window, win_length, n_fft)
    plan = get_plan(y.device, n_fft, hop_length, center, pad_mode if center else "constant",
                    win, mel_basis=basis, power=power,
                    dct_basis=dct_matrix_np(n_mfcc, n_mels, norm), lifter=lift)
    B, T = y.shape
    F = plan.num_frames(T)
    lib = _lib.lib()
    out = torch.empty((B, n_mfcc, F), dtype=torch.float32, device=y.device)
    ws_bytes = lib.mia_mfcc_workspace_bytes(plan.handle, B, T)
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=y.device)
    with torch.cuda.device(y.device):
        _lib.check(lib.mia_mfcc(plan.handle, y.data_ptr(), B, T, 80.0, int(bool(per_clip)),
                                out.data_ptr(), ws.data_ptr(), ws_bytes, stream_ptr(y.device)))
    return out[0] if is_1d else out
