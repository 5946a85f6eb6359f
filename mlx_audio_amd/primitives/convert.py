"""Decibel conversions (mirror of python/mlx_audio/primitives/convert.py)."""

from __future__ import annotations

import torch

from .. import _lib
from .._runtime import require_gpu, stream_ptr


def _to_db(S, ref, coefficient: float, amin: float, top_db, per_clip: bool) -> torch.Tensor:
    """coef*log10(max(S,amin)/max(ref,amin)) with the top_db floor (convert.py:14-57).

    The floor uses the maximum of the WHOLE array, as the reference does;
    `per_clip=True` takes it per leading-axis entry instead.
    """
    require_gpu()
    if not isinstance(S, torch.Tensor):
        S = torch.as_tensor(S)
    if not S.is_cuda:
        S = S.to("cuda")
    S = S.to(torch.float32).contiguous()
    if top_db is not None and top_db <= 0:
        raise ValueError(f"top_db must be positive, got {top_db}")
    if callable(ref):
        # the reference hands its array to the callable (convert.py:42-44); NumPy reducers such
        # as np.max - the usual idiom - do not take a device tensor, so they get a host copy
        try:
            ref_value = float(ref(S))
        except (TypeError, RuntimeError, ValueError):
            ref_value = float(ref(S.detach().cpu().numpy()))
    else:
        ref_value = float(ref)
    if per_clip and S.dim() >= 2:
        B, n = S.shape[0], S[0].numel()
    else:
        B, n = 1, S.numel()
    out = torch.empty_like(S)
    ws = torch.empty((8 * (B + 1),), dtype=torch.uint8, device=S.device)
    with torch.cuda.device(S.device):
        _lib.check(_lib.lib().mia_to_db(
            S.data_ptr(), B, n, ref_value, float(amin), float(coefficient),
            -1.0 if top_db is None else float(top_db), int(bool(per_clip)), out.data_ptr(),
            ws.data_ptr(), stream_ptr(S.device)))
    return out


def power_to_db(S, ref=1.0, amin: float = 1e-10, top_db=80.0, per_clip: bool = False):
    """10*log10(S/ref) (reference convert.py:60-94)."""
    return _to_db(S, ref, 10.0, amin, top_db, per_clip)


def amplitude_to_db(S, ref=1.0, amin: float = 1e-5, top_db=80.0, per_clip: bool = False):
    """20*log10(S/ref) (reference convert.py:129-163)."""
    return _to_db(S, ref, 20.0, amin, top_db, per_clip)


def _from_db(S_db, ref: float, divisor: float) -> torch.Tensor:
    require_gpu()
    if not isinstance(S_db, torch.Tensor):
        S_db = torch.as_tensor(S_db)
    if not S_db.is_cuda:
        S_db = S_db.to("cuda")
    S_db = S_db.to(torch.float32).contiguous()
    out = torch.empty_like(S_db)
    with torch.cuda.device(S_db.device):
        _lib.check(_lib.lib().mia_from_db(S_db.data_ptr(), S_db.numel(), float(ref), float(divisor),
                                          out.data_ptr(), stream_ptr(S_db.device)))
    return out


def db_to_power(S_db, ref: float = 1.0) -> torch.Tensor:
    """ref * 10^(S_db/10) (reference convert.py:97-126)."""
    return _from_db(S_db, ref, 10.0)


def db_to_amplitude(S_db, ref: float = 1.0) -> torch.Tensor:
    """ref * 10^(S_db/20) (reference convert.py:166-195)."""
    return _from_db(S_db, ref, 20.0)
