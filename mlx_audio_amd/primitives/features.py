"""Spectral features on MI355X with the signatures of `mlx_audio.primitives.features`
(reference python/mlx_audio/primitives/features.py): spectral_centroid, spectral_bandwidth,
spectral_rolloff, spectral_flatness, spectral_contrast, zero_crossing_rate.

Each statistic is one kernel launch over the (rows, bins, frames) spectrogram; from a waveform
the spectrogram comes from this package's `stft` + `magnitude`.
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._runtime import as_device_f32, get_plan, require_gpu, stream_ptr
from .stft import _resolve, magnitude, stft
from .windows import padded_window_np

_STAT = {"centroid": 0, "bandwidth": 1, "rolloff": 2, "flatness": 3}


def _get_frequencies(sr, n_fft: int) -> np.ndarray:
    """Bin centre frequencies (features.py:21-23)."""
    return np.linspace(0, sr / 2.0, n_fft // 2 + 1).astype(np.float32)


def _compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode):
    """Magnitude spectrogram unless one is given (features.py:26-56)."""
    if S is not None:
        require_gpu()
        if not isinstance(S, torch.Tensor):
            S = torch.as_tensor(np.asarray(S))
        if not S.is_cuda:
            S = S.to("cuda")
        return S.to(torch.float32)
    if y is None:
        raise ValueError("Either y (audio) or S (spectrogram) must be provided")
    return magnitude(stft(y, n_fft=n_fft, hop_length=hop_length, win_length=win_length,
                          window=window, center=center, pad_mode=pad_mode))


_freq_dev_cache: dict = {}


def _default_freq_dev(sr, n_fft: int, device) -> torch.Tensor:
    """Device copy of the default bin frequencies, kept per (sr, n_fft, device)."""
    key = (float(sr), int(n_fft), device.index)
    t = _freq_dev_cache.get(key)
    if t is None:
        if len(_freq_dev_cache) >= 64:
            _freq_dev_cache.clear()
        t = torch.from_numpy(_get_frequencies(sr, n_fft)).to(device)
        _freq_dev_cache[key] = t
    return t


def fused_statistics(y, sr: int = 22050, n_fft: int = 2048, hop_length: int = 512, win_length=None,
                     window="hann", center: bool = True, pad_mode: str = "constant", p: float = 2.0,
                     norm: bool = True, roll_percent: float = 0.85, power: float = 2.0,
                     amin: float = 1e-10):
    """Centroid, bandwidth, roll-off and flatness of a waveform in one launch: the forward STFT
    kernel with the statistics as its epilogue, so no spectrogram reaches memory.  Returns a
    (4, frames) or (batch, 4, frames) tensor in that order.  No counterpart function in the
    reference (its four functions each compute their own spectrogram, features.py:26-56); the
    four mirror functions below use it whenever they are called with a waveform."""
    hop_length, win_length = _resolve(n_fft, hop_length, win_length)
    if center and pad_mode not in _lib.PAD_MODES:
        raise ValueError(f"Unknown pad_mode: '{pad_mode}'. Supported: reflect, constant, edge")
    x = as_device_f32(y)
    is_1d = x.dim() == 1
    xb = x[None] if is_1d else x
    win = padded_window_np(window, win_length, n_fft)
    plan = get_plan(xb.device, n_fft, hop_length, center, pad_mode if center else "constant", win)
    B, T = xb.shape
    frames = plan.num_frames(T)
    freq = _default_freq_dev(sr, n_fft, xb.device)
    out = torch.empty((B, 4, frames), dtype=torch.float32, device=xb.device)
    with torch.cuda.device(xb.device):
        _lib.check(_lib.lib().mia_spectral_stats(
            plan.handle, xb.data_ptr(), B, T, freq.data_ptr(), float(p), int(bool(norm)),
            float(roll_percent), float(power), float(amin), out.data_ptr(), stream_ptr(xb.device)))
    return out[0] if is_1d else out


def spectrogram_statistics(S, sr: int = 22050, n_fft: int = 2048, freq=None, p: float = 2.0,
                           norm: bool = True, roll_percent: float = 0.85, amin: float = 1e-10,
                           power: float = 1.0) -> torch.Tensor:
    """Centroid, bandwidth, roll-off and flatness of a magnitude spectrogram in one launch that
    reads it twice (the four separate functions read it five times).  Returns (4, frames) or
    (batch, 4, frames); `power` is applied to S for the flatness only (2.0 reproduces
    spectral_flatness's default on magnitudes).  No counterpart function in the reference."""
    if roll_percent < 0.0 or roll_percent > 1.0:
        raise ValueError(f"roll_percent must be within [0, 1], got {roll_percent}")
    S = _compute_spectrogram(None, S, n_fft, 512, None, "hann", True, "constant")
    batched = S.dim() == 3
    Sb = (S if batched else S[None]).contiguous()
    rows, bins, frames = Sb.shape
    fdev = _default_freq_dev(sr, n_fft, Sb.device) if freq is None else _dev_vector(freq, Sb.device)
    if fdev.numel() != bins:
        raise ValueError(f"freq has {fdev.numel()} entries but the spectrogram has {bins} bins")
    out = torch.empty((rows, 4, frames), dtype=torch.float32, device=Sb.device)
    with torch.cuda.device(Sb.device):
        _lib.check(_lib.lib().mia_spectral_stat(
            Sb.data_ptr(), rows, bins, frames, fdev.data_ptr(), 0, 4, float(p), int(bool(norm)),
            float(roll_percent), float(amin), float(power), out.data_ptr(), stream_ptr(Sb.device)))
    return out if batched else out[0]


def _fused_ok(y, S, freq, n_fft) -> bool:
    return (S is None and y is not None and freq is None
            and bool(_lib.lib().mia_fft_size_supported(int(n_fft))))


def _pick(stats: torch.Tensor, which: int) -> torch.Tensor:
    return stats[..., which:which + 1, :]


def _dev_vector(v, device) -> torch.Tensor:
    if isinstance(v, torch.Tensor):
        return v.to(device=device, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(v, dtype=np.float32))).to(device)


def _stat(S: torch.Tensor, kind: str, freq=None, aux=None, p=2.0, norm=True, fraction=0.0,
          amin=0.0, power=1.0) -> torch.Tensor:
    """Runs one statistic over S (rows, bins, frames) or (bins, frames); keeps the reference's
    output shapes: (rows, 1, frames) or (1, frames)."""
    batched = S.dim() == 3
    Sb = (S if batched else S[None]).contiguous()
    if Sb.dim() != 3:
        raise ValueError(f"expected a 2-D or 3-D spectrogram, got {S.dim()}-D")
    rows, bins, frames = Sb.shape
    dev = Sb.device
    fdev = None
    if freq is not None:
        fdev = _dev_vector(freq, dev)
        if fdev.numel() != bins:
            raise ValueError(f"freq has {fdev.numel()} entries but the spectrogram has {bins} bins")
    adev = None
    if aux is not None:
        adev = _dev_vector(aux, dev).reshape(-1)
        if adev.numel() != rows * frames:
            raise ValueError("centroid must hold one value per frame")
    out = torch.empty((rows, 1, frames), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().mia_spectral_stat(
            Sb.data_ptr(), rows, bins, frames, 0 if fdev is None else fdev.data_ptr(),
            0 if adev is None else adev.data_ptr(), _STAT[kind], float(p), int(bool(norm)),
            float(fraction), float(amin), float(power), out.data_ptr(), stream_ptr(dev)))
    return out if batched else out[0]


def spectral_centroid(y=None, sr: int = 22050, S=None, n_fft: int = 2048, hop_length: int = 512,
                      win_length=None, window="hann", center: bool = True,
                      pad_mode: str = "constant", freq=None) -> torch.Tensor:
    """Centre of mass of the spectrum per frame (reference features.py:59-136)."""
    if _fused_ok(y, S, freq, n_fft):
        return _pick(fused_statistics(y, sr, n_fft, hop_length, win_length, window, center, pad_mode), 0)
    S = _compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode)
    if freq is None:
        freq = _default_freq_dev(sr, n_fft, S.device)
    return _stat(S, "centroid", freq=freq)


def spectral_bandwidth(y=None, sr: int = 22050, S=None, n_fft: int = 2048, hop_length: int = 512,
                       win_length=None, window="hann", center: bool = True,
                       pad_mode: str = "constant", freq=None, centroid=None, p: float = 2.0,
                       norm: bool = True) -> torch.Tensor:
    """Spread of the spectrum around its centroid (reference features.py:139-241)."""
    if centroid is None and _fused_ok(y, S, freq, n_fft):
        return _pick(fused_statistics(y, sr, n_fft, hop_length, win_length, window, center, pad_mode,
                                      p=p, norm=norm), 1)
    S = _compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode)
    if freq is None:
        freq = _default_freq_dev(sr, n_fft, S.device)
    if centroid is None:
        centroid = _stat(S, "centroid", freq=freq)
    return _stat(S, "bandwidth", freq=freq, aux=centroid, p=p, norm=norm)


def spectral_rolloff(y=None, sr: int = 22050, S=None, n_fft: int = 2048, hop_length: int = 512,
                     win_length=None, window="hann", center: bool = True,
                     pad_mode: str = "constant", freq=None, roll_percent: float = 0.85,
                     use_cpp: bool = True) -> torch.Tensor:
    """Frequency below which `roll_percent` of the energy lies (reference features.py:296-382).
    `use_cpp` is accepted for signature compatibility; there is one implementation here."""
    if roll_percent < 0.0:
        raise ValueError(f"roll_percent must be >= 0.0, got {roll_percent}")
    if roll_percent > 1.0:
        raise ValueError(f"roll_percent must be <= 1.0, got {roll_percent}")
    if _fused_ok(y, S, freq, n_fft):
        return _pick(fused_statistics(y, sr, n_fft, hop_length, win_length, window, center, pad_mode,
                                      roll_percent=roll_percent), 2)
    S = _compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode)
    if freq is None:
        freq = _default_freq_dev(sr, n_fft, S.device)
    return _stat(S, "rolloff", freq=freq, fraction=roll_percent)


def spectral_flatness(y=None, S=None, n_fft: int = 2048, hop_length: int = 512, win_length=None,
                      window="hann", center: bool = True, pad_mode: str = "constant",
                      power: float = 2.0, amin: float = 1e-10) -> torch.Tensor:
    """Wiener entropy per frame (reference features.py:385-464).  `power` applies to the
    magnitude computed from `y`; a spectrogram passed as `S` is used as it is."""
    if _fused_ok(y, S, None, n_fft):
        return _pick(fused_statistics(y, 22050, n_fft, hop_length, win_length, window, center, pad_mode,
                                      power=power, amin=amin), 3)
    from_y = S is None
    S = _compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode)
    return _stat(S, "flatness", amin=amin, power=power if from_y else 1.0)


_CONTRAST_MAX_Q = 64  # values per quantile the kernel's sorted register arrays hold


def _contrast_bands(freq: np.ndarray, fmin: float, n_bands: int, quantile: float) -> np.ndarray:
    """(n_bands + 1, 3) int32 rows {first bin, one past the last bin, values per quantile} of
    the octave bands [0, fmin], [fmin, 2 fmin], ... (reference features.py:566-599): a band takes
    the bins inside its edges plus the bin just below (not the first band), the last band runs to
    the Nyquist bin, the count per quantile is taken before the top bin is dropped (every band
    but the last drops it).  An empty band is {0, 0, 1}."""
    octa = np.zeros(n_bands + 2)
    octa[1:] = fmin * (2.0 ** np.arange(0, n_bands + 1))
    rows = []
    for k, (f_low, f_high) in enumerate(zip(octa[:-1], octa[1:])):
        mask = np.logical_and(freq >= f_low, freq <= f_high)
        idx = np.flatnonzero(mask)
        if len(idx) == 0:
            rows.append((0, 0, 1))
            continue
        if k > 0 and idx[0] > 0:
            mask[idx[0] - 1] = True
        if k == n_bands and idx[-1] + 1 < len(mask):
            mask[idx[-1] + 1:] = True
        sel = np.flatnonzero(mask)
        if not np.array_equal(sel, np.arange(sel[0], sel[-1] + 1)):
            raise NotImplementedError("spectral_contrast needs increasing bin frequencies")
        nq = int(np.maximum(np.rint(quantile * len(sel)), 1))
        hi = sel[-1] + 1
        if k < n_bands and len(sel) > 1:
            hi -= 1
        rows.append((int(sel[0]), int(hi), nq))
    return np.asarray(rows, dtype=np.int32)


def spectral_contrast(y=None, sr: int = 22050, S=None, n_fft: int = 2048, hop_length: int = 512,
                      win_length=None, window="hann", center: bool = True,
                      pad_mode: str = "constant", freq=None, fmin: float = 200.0,
                      n_bands: int = 6, quantile: float = 0.02,
                      linear: bool = False) -> torch.Tensor:
    """Peak-to-valley contrast per octave band and frame (reference features.py:467-639)."""
    if n_bands <= 0:
        raise ValueError(f"n_bands must be positive, got {n_bands}")
    if quantile < 0.0:
        raise ValueError(f"quantile must be >= 0.0, got {quantile}")
    if quantile > 1.0:
        raise ValueError(f"quantile must be <= 1.0, got {quantile}")
    S = _compute_spectrogram(y, S, n_fft, hop_length, win_length, window, center, pad_mode)
    if freq is None:
        freq_np = _get_frequencies(sr, n_fft)
    elif isinstance(freq, torch.Tensor):
        freq_np = freq.detach().cpu().numpy().astype(np.float32)
    else:
        freq_np = np.asarray(freq, dtype=np.float32)
    batched = S.dim() == 3
    Sb = (S if batched else S[None]).contiguous()
    rows, bins, frames = Sb.shape
    if freq_np.shape != (bins,):
        raise ValueError(f"freq has {freq_np.size} entries but the spectrogram has {bins} bins")
    bands = _contrast_bands(freq_np, float(fmin), int(n_bands), float(quantile))
    live = bands[:, 1] - bands[:, 0] > bands[:, 2]   # bands that actually select quantiles
    # The kernel keeps at most 64 values per quantile in registers.  Bands that need more (wide
    # top bands with a large quantile, or very long transforms) are rare: they are handed to the
    # kernel as empty bands and filled in below from a device sort of just their bins.
    wide = live & (bands[:, 2] > _CONTRAST_MAX_Q)
    kbands = bands.copy()
    kbands[wide] = (0, 0, 1)
    max_q = int(kbands[live & ~wide, 2].max()) if (live & ~wide).any() else 1
    bands_dev = torch.from_numpy(kbands).to(Sb.device)
    out = torch.empty((rows, n_bands + 1, frames), dtype=torch.float32, device=Sb.device)
    with torch.cuda.device(Sb.device):
        _lib.check(_lib.lib().mia_spectral_contrast(
            Sb.data_ptr(), rows, bins, frames, bands_dev.data_ptr(), n_bands + 1, max_q,
            int(bool(linear)), out.data_ptr(), stream_ptr(Sb.device)))
    for k in np.flatnonzero(wide):
        lo, hi, nq = (int(v) for v in bands[k])
        band = torch.sort(Sb[:, lo:hi, :], dim=1).values
        valley = band[:, :nq, :].mean(dim=1)
        peak = band[:, hi - lo - nq:, :].mean(dim=1)
        if linear:
            out[:, k, :] = peak - valley
        else:
            out[:, k, :] = (10.0 * torch.log10(torch.clamp(peak, min=1e-10))
                            - 10.0 * torch.log10(torch.clamp(valley, min=1e-10)))
    return out if batched else out[0]


def zero_crossing_rate(y, frame_length: int = 2048, hop_length: int = 512, center: bool = True,
                       pad_mode: str = "edge", use_mlx: bool = True) -> torch.Tensor:
    """Fraction of sign changes per frame (reference features.py:672-769).  `use_mlx=True`
    takes the sign as x >= 0, `False` as the IEEE sign bit (they differ for -0.0 and NaN)."""
    if frame_length <= 0:
        raise ValueError(f"frame_length must be positive, got {frame_length}")
    if hop_length <= 0:
        raise ValueError(f"hop_length must be positive, got {hop_length}")
    if center and pad_mode not in ("constant", "edge"):
        raise ValueError(f"Unknown pad_mode: '{pad_mode}'. Supported: 'constant', 'edge'")
    x = as_device_f32(y)
    is_1d = x.dim() == 1
    xb = x[None] if is_1d else x
    rows, n = xb.shape
    padded = n + (2 * (frame_length // 2) if center else 0)
    if padded < frame_length:
        raise ValueError(
            f"Input signal length ({padded}) must be >= frame_length ({frame_length})")
    frames = 1 + (padded - frame_length) // hop_length
    out = torch.empty((rows, 1, frames), dtype=torch.float32, device=xb.device)
    with torch.cuda.device(xb.device):
        _lib.check(_lib.lib().mia_zero_crossing_rate(
            xb.data_ptr(), rows, n, frame_length, hop_length, int(bool(center)),
            int(pad_mode == "edge"), int(not use_mlx), out.data_ptr(), stream_ptr(xb.device)))
    return out[0] if is_1d else out
