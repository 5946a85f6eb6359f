"""Griffin-Lim phase reconstruction on MI355X with the signatures of
`mlx_audio.primitives.griffinlim` (reference python/mlx_audio/primitives/griffinlim.py).

An iteration is three launches: the fused inverse STFT, the fused forward STFT and one
elementwise update (phase of the new spectrum, target magnitudes, momentum).  The complex
spectra stay in the (rows, frames, bins) order the STFT kernels write and read, so nothing is
transposed inside the loop.
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._runtime import require_gpu, stream_ptr
from .stft import _istft_raw, _stft_raw, istft, magnitude, phase, stft


_FEED_UNTRIMMED = True   # the loop's shortcut described in griffinlim(); tests switch it off


def _dev_f32(x) -> torch.Tensor:
    require_gpu()
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x))
    if not x.is_cuda:
        x = x.to("cuda")
    return x.to(torch.float32)


def _polar(mag: torch.Tensor, angle: torch.Tensor) -> torch.Tensor:
    """mag * exp(i angle), elementwise over two contiguous tensors of one shape."""
    mag = mag.contiguous()
    angle = angle.contiguous()
    out = torch.empty(mag.shape, dtype=torch.complex64, device=mag.device)
    with torch.cuda.device(mag.device):
        _lib.check(_lib.lib().mia_polar(mag.data_ptr(), angle.data_ptr(), mag.numel(),
                                        out.data_ptr(), stream_ptr(mag.device)))
    return out


def _fit_frames(C: torch.Tensor, n_frames: int) -> torch.Tensor:
    """C: (rows, frames', bins) -> exactly n_frames frames (cut, or zero frames appended)."""
    if C.shape[1] > n_frames:
        C = C[:, :n_frames]
    elif C.shape[1] < n_frames:
        C = torch.nn.functional.pad(C, (0, 0, 0, n_frames - C.shape[1]))
    return C.contiguous()


def _update(c_new, mag_t, tprev, rebuilt, momentum: float) -> None:
    with torch.cuda.device(mag_t.device):
        _lib.check(_lib.lib().mia_griffinlim_update(
            c_new.data_ptr(), mag_t.data_ptr(), tprev.data_ptr(), rebuilt.data_ptr(),
            mag_t.numel(), float(momentum), stream_ptr(mag_t.device)))


def griffinlim(S, n_iter: int = 32, hop_length=None, win_length=None, n_fft=None, window="hann",
               center: bool = True, length=None, pad_mode: str = "constant",
               momentum: float = 0.99, init: str = "random", random_state=None) -> torch.Tensor:
    """Audio from a magnitude spectrogram by iterated phase estimation (reference
    griffinlim.py:17-196).  The random initial phase is drawn on the host with NumPy's
    generator, as the reference draws it, so a seed reproduces the reference's start."""
    if n_iter <= 0:
        raise ValueError(f"n_iter must be positive, got {n_iter}")
    if momentum < 0.0:
        raise ValueError(f"momentum must be >= 0.0, got {momentum}")
    if momentum >= 1.0:
        raise ValueError(f"momentum must be < 1.0, got {momentum}")
    S = _dev_f32(S)
    batched = S.dim() == 3
    if not batched:
        S = S[None]
    rows, bins, frames = S.shape
    if n_fft is None:
        n_fft = 2 * (bins - 1)
    if hop_length is None:
        hop_length = n_fft // 4
    if win_length is None:
        win_length = n_fft
    rng = np.random.default_rng(random_state)
    if init == "random":
        ang = rng.uniform(-np.pi, np.pi, (rows, bins, frames)).astype(np.float32)
        angles = torch.from_numpy(ang).to(S.device)
    elif init == "zeros":
        angles = None
    else:
        raise ValueError(f"Unknown init: '{init}'. Supported: 'random', 'zeros'")

    # everything below lives in (rows, frames, bins) order: one pass transposes the magnitudes and
    # builds the first estimate (and its copy, the "previous" one of the momentum term)
    Sc = S.contiguous()
    mag_t = torch.empty((rows, frames, bins), dtype=torch.float32, device=S.device)
    rebuilt = torch.empty((rows, frames, bins), dtype=torch.complex64, device=S.device)
    tprev = torch.empty_like(rebuilt)
    with torch.cuda.device(S.device):
        _lib.check(_lib.lib().mia_griffinlim_init(Sc.data_ptr(), 0 if angles is None else angles.data_ptr(),
                                                  rows, bins, frames, mag_t.data_ptr(), rebuilt.data_ptr(),
                                                  tprev.data_ptr(), stream_ptr(S.device)))
    kw = dict(hop_length=hop_length, win_length=win_length, n_fft=n_fft, window=window,
              center=center)
    # With centre padding by zeros the trimmed signal, padded again, is the untrimmed
    # overlap-add output with its first and last n_fft/2 samples zeroed: the forward transform
    # can read that buffer as it stands (no trimming view, no copy to make it contiguous) and
    # produces bit for bit what istft() followed by stft() does.
    in_place = _FEED_UNTRIMMED and center and pad_mode == "constant" and n_fft % 2 == 0
    for _ in range(n_iter):
        if in_place:
            padded = (length + n_fft) if length is not None else n_fft + (frames - 1) * hop_length
            y_full = _istft_raw(rebuilt.transpose(1, 2), n_fft, hop_length, win_length, window, padded)
            y_full[:, :n_fft // 2].zero_()
            y_full[:, padded - n_fft // 2:].zero_()
            c_new = _stft_raw(y_full, n_fft, hop_length, win_length, window, False, "constant")
            c_new = _fit_frames(c_new, frames)
        else:
            y = istft(rebuilt.transpose(1, 2), length=length, **kw)
            c_new = stft(y, pad_mode=pad_mode, **kw)            # (rows, bins, frames') view
            c_new = _fit_frames(c_new.transpose(1, 2), frames)  # physical order, exact frame count
        _update(c_new, mag_t, tprev, rebuilt, momentum)
    y = istft(rebuilt.transpose(1, 2), length=length, **kw)
    return y if batched else y[0]


def griffinlim_iter(S, angles, hop_length: int, win_length: int, n_fft: int, window="hann",
                    center: bool = True, pad_mode: str = "constant", momentum: float = 0.99,
                    tprev=None):
    """One Griffin-Lim iteration (reference griffinlim.py:199-284): returns
    (new_angles, new_rebuilt, error), error = mean((S - |STFT(ISTFT(S e^{i angles}))|)^2)."""
    S = _dev_f32(S)
    angles = _dev_f32(angles)
    rebuilt = _polar(S, angles)
    kw = dict(hop_length=hop_length, win_length=win_length, n_fft=n_fft, window=window,
              center=center)
    y = istft(rebuilt, **kw)
    c_new = stft(y, pad_mode=pad_mode, **kw)
    mag_new = magnitude(c_new)
    error = torch.mean((S - mag_new) ** 2)   # a monitoring scalar: one torch reduction
    new_angles = phase(c_new)
    c_cont = c_new.contiguous()
    prev = torch.zeros_like(c_cont) if tprev is None else tprev.to(torch.complex64).contiguous().clone()
    out = torch.empty_like(c_cont)
    use_m = momentum if (momentum > 0 and tprev is not None) else 0.0
    _update(c_cont, S.contiguous(), prev, out, use_m)
    return new_angles, out, error
