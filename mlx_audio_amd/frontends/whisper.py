"""Whisper log-mel front-end (mirror of
python/mlx_audio/models/whisper/inference.py:78-136)."""

from __future__ import annotations

import torch

from .. import _lib
from .._runtime import as_device_f32
from ..primitives.mel import _mel_raw

# reference constants/audio_processing.py:10-13, constants/spectral.py:12-26
WHISPER_N_FFT = 400
WHISPER_HOP_LENGTH = 160
WHISPER_SAMPLE_RATE = 16000
WHISPER_FMAX = 8000.0
WHISPER_N_SAMPLES = 480000


def compute_log_mel_spectrogram(audio, n_mels: int = 80, n_fft: int = WHISPER_N_FFT,
                                hop_length: int = WHISPER_HOP_LENGTH,
                                sample_rate: int = WHISPER_SAMPLE_RATE,
                                per_clip: bool = False) -> torch.Tensor:
    """Log-mel spectrogram with Whisper's normalisation.

    audio [T] or [B, T] -> [n_mels, frames] or [B, n_mels, frames]:
    melspectrogram(fmin=0, fmax=8000) -> log10(max(., 1e-10)) ->
    max(., max - 8) -> (. + 4) / 4.  As in the reference the maximum spans the
    whole array given (for a [B, T] batch: the batch); `per_clip=True` takes it
    per clip, which equals calling the function clip by clip as the reference's
    own transcription loop does.  One fused kernel plus a floor pass that only
    touches tiles holding values below the floor.
    """
    y = as_device_f32(audio)
    is_1d = y.dim() == 1
    if is_1d:
        y = y[None, :]
    out = _mel_raw(y, sample_rate, n_fft, hop_length, None, "hann", True, "constant", 2.0,
                   n_mels, 0.0, WHISPER_FMAX, False, "slaney", epilogue=_lib.EPI_WHISPER,
                   per_clip=per_clip)
    return out[0] if is_1d else out


def pad_or_trim(audio: torch.Tensor, length: int = WHISPER_N_SAMPLES) -> torch.Tensor:
    """Zero-pad or cut the last axis to `length` samples (inference.py:118-136)."""
    n = audio.shape[-1]
    if n > length:
        return audio[..., :length]
    if n < length:
        return torch.nn.functional.pad(audio, (0, length - n))
    return audio
