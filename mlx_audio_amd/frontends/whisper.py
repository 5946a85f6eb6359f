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


def sinusoids(length: int, dim: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """Sinusoidal positional embedding [length, dim] of the audio encoder (encoder.py:16-35):
    sin of position * inv_timescale in the first dim // 2 columns, cos in the rest."""
    import math

    half = dim // 2
    log_timescale = math.log(max_timescale) / (half - 1)
    inv = torch.exp(-log_timescale * torch.arange(half, dtype=torch.float32))
    scaled = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=1)


class WhisperEncoderStem:
    """The two convolutions in front of Whisper's encoder blocks (mirror of
    `AudioEncoder.__init__/__call__`, encoder.py:72-109): conv1 (n_mels -> n_state, kernel 3,
    padding 1) -> GELU -> conv2 (n_state -> n_state, kernel 3, stride 2, padding 1) -> GELU ->
    + positional_embedding[:frames'].

    Weights come in the reference's layout, (out_channels, kernel, in_channels), biases
    (out_channels,).  `precision` chooses the arithmetic of the convolutions:

    * "f32": weights, spectrogram and the intermediate stay float32 and the products run on the
      float32 matrix instruction (a chain of fused multiply-adds in float32) - what the
      reference's `nn.Conv1d` computes on a float32 checkpoint;
    * "f16": weights, spectrogram and the conv1 -> GELU intermediate are rounded to float16,
      accumulation in float32 - the precision of a float16 checkpoint, at 16 x the rate;
    * None (default): "f16" when the weights arrive as float16 tensors, "f32" otherwise, so a
      float32 checkpoint is never narrowed without the caller asking for it.

    Bias, GELU (exact, erf form) and the embedding are float32 either way.
    """

    def __init__(self, conv1_weight, conv1_bias, conv2_weight, conv2_bias, positional_embedding=None,
                 precision=None):
        if precision is None:
            dts = [getattr(w, "dtype", None) for w in (conv1_weight, conv2_weight)]
            half = all(str(d) in ("torch.float16", "float16") for d in dts)
            precision = "f16" if half else "f32"
        if precision not in ("f16", "f32"):
            raise ValueError(f"precision must be 'f16', 'f32' or None, got {precision!r}")
        self.precision = precision
        self._pcode = _lib.STEM_F16 if precision == "f16" else _lib.STEM_F32
        w1 = as_device_f32(conv1_weight, ndim_allowed=(3,))
        w2 = as_device_f32(conv2_weight, ndim_allowed=(3,))
        n_state, k1, n_mels = w1.shape
        if k1 != 3 or tuple(w2.shape) != (n_state, 3, n_state):
            raise ValueError("expected conv1 weight (n_state, 3, n_mels) and conv2 weight (n_state, 3, n_state), "
                             f"got {tuple(w1.shape)} and {tuple(w2.shape)}")
        if n_state % 128 != 0:
            raise NotImplementedError(f"n_state must be a multiple of 128 (got {n_state})")
        self.n_mels, self.n_state = int(n_mels), int(n_state)
        self.device = w1.device
        self.bias1 = as_device_f32(conv1_bias, ndim_allowed=(1,)).to(self.device)
        self.bias2 = as_device_f32(conv2_bias, ndim_allowed=(1,)).to(self.device)
        if self.bias1.numel() != n_state or self.bias2.numel() != n_state:
            raise ValueError("biases must have n_state elements")
        self.pos = None
        if positional_embedding is not None:
            pos = as_device_f32(positional_embedding, ndim_allowed=(2,)).to(self.device)
            if pos.shape[1] != n_state:
                raise ValueError("positional_embedding must be (n_ctx, n_state)")
            self.pos = pos
        lib = _lib.lib()
        nbytes = lib.mia_whisper_stem_weight_bytes(self.n_mels, self.n_state, self._pcode)
        self.packed = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.mia_whisper_stem_pack(w1.data_ptr(), w2.data_ptr(), self.n_mels, self.n_state,
                                                 self._pcode, self.packed.data_ptr(),
                                                 torch.cuda.current_stream(self.device).cuda_stream))
        self._ws = None

    def __call__(self, mel, out_dtype=None) -> torch.Tensor:
        """mel [B, n_mels, T] or [n_mels, T] -> [B, (T - 1) // 2 + 1, n_state]; out_dtype
        defaults to the precision's own type (float32 for "f32", float16 for "f16")."""
        if out_dtype is None:
            out_dtype = torch.float32 if self.precision == "f32" else torch.float16
        x = as_device_f32(mel, ndim_allowed=(2, 3)).to(self.device)
        if x.dim() == 2:
            x = x[None]
        B, C, T = x.shape
        if C != self.n_mels:
            raise ValueError(f"expected {self.n_mels} mel bands, got {C}")
        if T < 1:
            raise ValueError("empty spectrogram")
        if out_dtype not in (torch.float16, torch.float32):
            raise ValueError("out_dtype must be torch.float16 or torch.float32")
        t_out = (T - 1) // 2 + 1
        if self.pos is not None and self.pos.shape[0] < t_out:
            raise ValueError(f"positional_embedding holds {self.pos.shape[0]} positions, {t_out} needed")
        lib = _lib.lib()
        out = torch.empty((B, t_out, self.n_state), dtype=out_dtype, device=self.device)
        if B == 0:
            return out
        # clips per call: the kernel indexes frames with 32 bits
        per_call = max(1, min(B, ((1 << 31) - 512) // T))
        need = lib.mia_whisper_stem_workspace_bytes(per_call, T, self.n_mels, self.n_state, self._pcode)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            for b0 in range(0, B, per_call):
                nb = min(per_call, B - b0)
                _lib.check(lib.mia_whisper_stem(
                    x[b0:b0 + nb].data_ptr(), nb, T, self.n_mels, self.n_state, self._pcode,
                    self.packed.data_ptr(),
                    self.bias1.data_ptr(), self.bias2.data_ptr(),
                    self.pos.data_ptr() if self.pos is not None else None,
                    out[b0:b0 + nb].data_ptr(), 1 if out_dtype == torch.float16 else 0,
                    self._ws.data_ptr(), self._ws.numel(), stream))
        return out


def whisper_encoder_stem(mel, conv1_weight, conv1_bias, conv2_weight, conv2_bias, positional_embedding=None,
                         out_dtype=None, precision=None) -> torch.Tensor:
    """One-shot form of `WhisperEncoderStem` (packs the weights on every call)."""
    return WhisperEncoderStem(conv1_weight, conv1_bias, conv2_weight, conv2_bias, positional_embedding,
                              precision=precision)(mel, out_dtype=out_dtype)
