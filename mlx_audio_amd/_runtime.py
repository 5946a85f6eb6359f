"""Device plumbing shared by the primitives: tensor coercion, streams, plans.

PyTorch-ROCm supplies device memory and streams only; all arithmetic on the
hot path happens in libmia_hip.so.
"""

from __future__ import annotations

import ctypes
import hashlib
import threading
from collections import OrderedDict

import numpy as np
import torch

from . import _lib

_PLAN_CACHE_MAX = 64  # the reference bounds its window / filterbank caches too


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise RuntimeError(
            "mlx_audio_amd needs a ROCm GPU: its spectral kernels have no CPU fallback."
        )


def as_device_f32(x, ndim_allowed=(1, 2)) -> torch.Tensor:
    """Coerce array-like input to a contiguous float32 tensor on a GPU."""
    require_gpu()
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x))
    if x.dim() not in ndim_allowed:
        raise ValueError(f"expected input with ndim in {ndim_allowed}, got {x.dim()}D")
    if not x.is_cuda:
        x = x.to("cuda")
    if x.dtype != torch.float32:
        x = x.to(torch.float32)
    return x.contiguous()


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# Digests of the table arrays.  The mirror's table builders are lru-cached as immutable
# `bytes` and handed out as read-only views of them, so the digest is remembered per bytes
# object (a Whisper call would otherwise hash ~100 KB of filterbank every time); any other
# array is hashed on every call.
_digests: "OrderedDict[int, tuple[bytes, str]]" = OrderedDict()


def _digest(a: np.ndarray | None) -> str:
    if a is None:
        return "-"
    base = a
    while isinstance(base, np.ndarray) and base.base is not None:
        base = base.base
    whole = isinstance(base, bytes) and a.nbytes == len(base) and a.flags.c_contiguous
    if whole:
        hit = _digests.get(id(base))
        if hit is not None and hit[0] is base:
            return hit[1]
    d = hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
    if whole:
        _digests[id(base)] = (base, d)  # holds the bytes, so the id cannot be reused
        while len(_digests) > 256:
            _digests.popitem(last=False)
    return d


def _fptr(a: np.ndarray | None):
    if a is None:
        return ctypes.POINTER(ctypes.c_float)()
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


class Plan:
    """Owns one `mia_plan` (device tables for one parameter set on one GPU)."""

    def __init__(self, device, n_fft, hop, center, pad_mode, window, mel_basis, power,
                 dct_basis, lifter):
        lib = _lib.lib()
        window = np.ascontiguousarray(window, dtype=np.float32)
        if window.shape != (n_fft,):
            raise ValueError(f"Window array length ({window.shape[0]}) must match n_fft ({n_fft})")
        if mel_basis is not None:
            mel_basis = np.ascontiguousarray(mel_basis, dtype=np.float32)
        if dct_basis is not None:
            dct_basis = np.ascontiguousarray(dct_basis, dtype=np.float32)
        if lifter is not None:
            lifter = np.ascontiguousarray(lifter, dtype=np.float32)
        desc = _lib.PlanDesc(
            n_fft=n_fft, hop_length=hop, center=int(bool(center)),
            pad_mode=_lib.PAD_MODES[pad_mode], window=_fptr(window),
            n_mels=0 if mel_basis is None else mel_basis.shape[0], mel_basis=_fptr(mel_basis),
            power=float(power), n_mfcc=0 if dct_basis is None else dct_basis.shape[0],
            dct_basis=_fptr(dct_basis), lifter=_fptr(lifter),
        )
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.mia_plan_create(ctypes.byref(desc), ctypes.byref(handle)))
        self.handle = handle
        self.device = device
        self.n_fft = n_fft
        self.hop = hop
        self.n_mels = desc.n_mels
        self.n_mfcc = desc.n_mfcc

    def num_frames(self, T: int) -> int:
        n = _lib.lib().mia_plan_num_frames(self.handle, int(T))
        if n < 0:
            _lib.check(int(n))
        return int(n)

    def close(self) -> None:
        if self.handle:
            try:
                with torch.cuda.device(self.device):
                    _lib.lib().mia_plan_destroy(self.handle)
            finally:
                self.handle = None

    def __del__(self):
        # The plan is destroyed when its last holder lets go of it: the cache only drops its
        # reference on eviction, so a caller that still works with an evicted plan keeps it alive.
        try:
            self.close()
        except Exception:
            pass


_plans: "OrderedDict[tuple, Plan]" = OrderedDict()
_plans_lock = threading.Lock()


def get_plan(device: torch.device, n_fft: int, hop: int, center: bool, pad_mode: str,
             window: np.ndarray, mel_basis: np.ndarray | None = None, power: float = 2.0,
             dct_basis: np.ndarray | None = None, lifter: np.ndarray | None = None) -> Plan:
    """LRU cache of plans keyed by every parameter that shapes the tables."""
    if pad_mode not in _lib.PAD_MODES:
        raise ValueError(f"Unknown pad_mode: '{pad_mode}'. Supported: reflect, constant, edge")
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev_index, n_fft, hop, bool(center), pad_mode, _digest(window), _digest(mel_basis),
           float(power), _digest(dct_basis), _digest(lifter))
    with _plans_lock:
        plan = _plans.get(key)
        if plan is not None:
            _plans.move_to_end(key)
            return plan
        plan = Plan(torch.device("cuda", dev_index), n_fft, hop, center, pad_mode, window,
                    mel_basis, power, dct_basis, lifter)
        _plans[key] = plan
        while len(_plans) > _PLAN_CACHE_MAX:
            _plans.popitem(last=False)  # freed by Plan.__del__ once no caller holds it
        return plan


def clear_plans() -> None:
    with _plans_lock:
        _plans.clear()
