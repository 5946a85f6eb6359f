"""HTDemucs spectrogram front-end (mirror of
python/mlx_audio/models/demucs/model.py:378-506)."""

from __future__ import annotations

import math

import torch

from .. import _lib
from .._runtime import as_device_f32, get_plan, stream_ptr
from ..primitives.stft import _istft_raw, _stft_raw
from ..primitives.windows import padded_window_np

DEMUCS_N_FFT = 4096      # reference constants/audio_processing.py:15-17
DEMUCS_HOP_LENGTH = 1024


def htdemucs_spec(x, n_fft: int = DEMUCS_N_FFT, hop_length: int = DEMUCS_HOP_LENGTH):
    """`HTDemucs._compute_stft` (model.py:378-417).

    x [B, C, T] -> complex64 [B, C, n_fft//2, ceil(T/hop)]: zero-pad like
    PyTorch demucs, reflect-padded STFT scaled by 1/sqrt(n_fft) (fused into the
    kernel's store), Nyquist bin dropped, frames [2 : 2+le] kept.  The result
    is a strided view of the kernel's (B*C, frames, bins) output.
    """
    return _spec_and_sums(x, n_fft, hop_length, False)[0]


def _spec_and_sums(x, n_fft: int, hop_length: int, want_sums: bool):
    """htdemucs_spec, and with want_sums (compiled FFT sizes) the per-item sums the input
    normalisation needs - sum and sum of squares of the real and imaginary parts of the kept
    bins and frames - accumulated by the STFT kernel while the spectrum is still in registers.
    Returns (spec, sums or None, sums of the waveform or None)."""
    x = as_device_f32(x, ndim_allowed=(3,))
    B, C, T = x.shape
    le = int(math.ceil(T / hop_length))
    pad = hop_length // 2 * 3
    right = pad + le * hop_length - T
    xp = torch.empty((B, C, pad + T + right), dtype=x.dtype, device=x.device)
    lib = _lib.lib()
    mix_sums = None
    if want_sums:
        # the padded copy also adds up the waveform's own moments (its normalisation needs them)
        mix_sums = torch.empty((B, 2), dtype=torch.float64, device=x.device)
        xc = x.contiguous()
        with torch.cuda.device(x.device):
            _lib.check(lib.mia_pad_rows_moments(xc.data_ptr(), B * C, T, pad, right, C, xp.data_ptr(),
                                                mix_sums.data_ptr(), stream_ptr(x.device)))
    else:
        # the zero margins written by themselves: a pad() would first fill the whole buffer
        xp[..., :pad].zero_()
        xp[..., pad + T:].zero_()
        xp[..., pad:pad + T].copy_(x)
    scale = 1.0 / math.sqrt(n_fft)
    sums = None
    if want_sums and lib.mia_fft_size_supported(n_fft):
        rows = xp.reshape(B * C, -1)
        plan = get_plan(x.device, n_fft, hop_length, True, "reflect", padded_window_np("hann", n_fft, n_fft))
        frames = plan.num_frames(rows.shape[1])
        raw = torch.empty((B * C, frames, n_fft // 2 + 1), dtype=torch.complex64, device=x.device)
        sums = torch.empty((B, 2), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.mia_stft_moments(plan.handle, rows.data_ptr(), B * C, rows.shape[1], scale,
                                            raw.data_ptr(), C, n_fft // 2, 2, 2 + le, sums.data_ptr(),
                                            stream_ptr(x.device)))
    else:
        raw = _stft_raw(xp.reshape(B * C, -1), n_fft, hop_length, n_fft, "hann", True, "reflect",
                        scale=scale)                       # (B*C, frames, bins)
    spec = raw.transpose(1, 2)[:, :-1, 2:2 + le]           # (B*C, n_fft//2, le)
    return spec.unflatten(0, (B, C)), sums, mix_sums


def htdemucs_ispec(m, length: int, n_fft: int = DEMUCS_N_FFT,
                   hop_length: int = DEMUCS_HOP_LENGTH):
    """`HTDemucs._mask_and_istft`, complex-as-channels branch (model.py:432-506).

    m real [B, S, 2C, F, Tf] laid out (re_ch0, im_ch0, re_ch1, ...) ->
    audio [B, S, C, length].  The sqrt(n_fft) factor is fused into the
    overlap-add store.
    """
    m = as_device_f32(m, ndim_allowed=(5,))
    B, S, C2, F, Tf = m.shape
    C = C2 // 2
    # one zero Nyquist bin, two zero frames each side, real / imaginary planes interleaved, and
    # the (rows, frames, bins) order the inverse kernel reads in whole rows: one launch
    planes = m.contiguous()        # [B, S, C, 2, F, Tf] as it stands
    z = torch.empty((B * S * C, Tf + 4, F + 1), dtype=torch.complex64, device=m.device)
    with torch.cuda.device(m.device):
        _lib.check(_lib.lib().mia_cac_to_spectrum(planes.data_ptr(), B * S * C, F, Tf, 1, 2,
                                                  z.data_ptr(), stream_ptr(m.device)))
    z = z.transpose(1, 2)          # (rows, F + 1, Tf + 4) view
    pad = hop_length // 2 * 3
    le = hop_length * int(math.ceil(length / hop_length)) + 2 * pad
    y = _istft_raw(z, n_fft, hop_length, n_fft, "hann", le + n_fft, scale=math.sqrt(n_fft))
    y = y[:, n_fft // 2:n_fft // 2 + le]
    return y.reshape(B, S, C, -1)[:, :, :, pad:pad + length]


def htdemucs_normalize(mix, spec=None, eps: float = 1e-5):
    """Input normalisation of `HTDemucs.__call__` (model.py:206-233).

    mix [B, C, T] and its spectrogram [B, C, F, Tf] (complex64, any strides; computed with
    `htdemucs_spec` if not given) -> (freq_in, spec_mean, spec_std, mix_norm, mix_mean, mix_std):
    freq_in [B, 2C, F, Tf] holds the real and imaginary parts as planes 2c, 2c+1, normalised by
    the mean and standard deviation over all of a batch item's planes; mix_norm [B, C, T]
    likewise over the item's channels and samples.  The std values include the +eps, as in the
    reference, where they are used later to undo the normalisation (model.py:349, 367).
    """
    mix = as_device_f32(mix, ndim_allowed=(3,))
    spec_sums = mix_sums = None
    if spec is None:
        spec, spec_sums, mix_sums = _spec_and_sums(mix, DEMUCS_N_FFT, DEMUCS_HOP_LENGTH, True)
    if spec.dim() != 4 or spec.dtype != torch.complex64:
        raise ValueError("spec must be a complex64 [B, C, F, Tf] tensor")
    B, C, T = mix.shape
    _, Cs, F, Tf = spec.shape
    dev = mix.device
    lib = _lib.lib()
    freq_in = torch.empty((B, 2 * Cs, F, Tf), dtype=torch.float32, device=dev)
    mix_norm = torch.empty_like(mix)
    s_stats = torch.empty((B, 2), dtype=torch.float32, device=dev)
    m_stats = torch.empty((B, 2), dtype=torch.float32, device=dev)
    sums = torch.empty((B, 2), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        stream = stream_ptr(dev)
        sb, sc, sf, st = spec.stride()
        base = spec.data_ptr()
        if spec_sums is not None:   # the STFT kernel already added the item's sums up
            _lib.check(lib.mia_normalize_cac_sums(base, B, Cs, F, Tf, sb, sc, sf, st, float(eps),
                                                  freq_in.data_ptr(), s_stats.data_ptr(),
                                                  spec_sums.data_ptr(), stream))
        else:
            _lib.check(lib.mia_normalize_cac(base, B, Cs, F, Tf, sb, sc, sf, st, float(eps),
                                             freq_in.data_ptr(), s_stats.data_ptr(), sums.data_ptr(),
                                             stream))
        if mix_sums is not None:    # added up while the waveform was padded for the STFT
            _lib.check(lib.mia_normalize_rows_sums(mix.data_ptr(), B, C * T, float(eps), mix_norm.data_ptr(),
                                                   m_stats.data_ptr(), mix_sums.data_ptr(), stream))
        else:
            _lib.check(lib.mia_normalize_rows(mix.data_ptr(), B, C * T, float(eps), mix_norm.data_ptr(),
                                              m_stats.data_ptr(), sums.data_ptr(), stream))
    return (freq_in, s_stats[:, 0].reshape(B, 1, 1, 1), s_stats[:, 1].reshape(B, 1, 1, 1),
            mix_norm, m_stats[:, 0].reshape(B, 1, 1), m_stats[:, 1].reshape(B, 1, 1))
