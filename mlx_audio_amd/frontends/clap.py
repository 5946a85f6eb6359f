"""CLAP log-mel features (mirror of
python/mlx_audio/models/clap/feature_extractor.py:193-277).

The reference computes these with librosa on the host; the same arithmetic is
melspectrogram(sr=48000, n_fft=1024, hop=480, n_mels=64, fmin=50, fmax=14000,
htk=True) -> power_to_db(ref=1, amin=1e-10, top_db=None) -> [T, F], which here
is one fused kernel over the whole batch.
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._runtime import as_device_f32
from ..primitives.mel import _mel_raw

CLAP_SAMPLE_RATE = 48000   # reference constants/audio_processing.py:19-24
CLAP_N_FFT = 1024
CLAP_HOP_LENGTH = 480
CLAP_N_MELS = 64
CLAP_FMIN = 50.0
CLAP_FMAX = 14000.0


def clap_log_mel(audio, sample_rate: int = CLAP_SAMPLE_RATE, n_fft: int = CLAP_N_FFT,
                 hop_length: int = CLAP_HOP_LENGTH, n_mels: int = CLAP_N_MELS) -> torch.Tensor:
    """`_compute_mel` (feature_extractor.py:193-212) for [T] or [B, T] input.

    Returns dB mel features [T_frames, n_mels] / [B, T_frames, n_mels], the reference's
    layout, written that way by the kernel (contiguous).
    """
    y = as_device_f32(audio)
    is_1d = y.dim() == 1
    if is_1d:
        y = y[None, :]
    out = _mel_raw(y, sample_rate, n_fft, hop_length, None, "hann", True, "constant", 2.0,
                   n_mels, CLAP_FMIN, CLAP_FMAX, True, "slaney", epilogue=_lib.EPI_DB, ref=1.0,
                   amin=1e-10, coef=10.0, top_db=-1.0, frames_first=True)
    return out[0] if is_1d else out


class CLAPFeatureExtractor:
    """Batch feature extractor with the interface of the reference class
    (feature_extractor.py:17-97), fallback (non-HuggingFace) semantics.

    Clips no longer than `max_length_s` are tiled / zero-padded to
    nb_max_samples and their mel is repeated on 4 channels (:214-231, :176-177).
    Longer clips get the 4-channel "fusion" stack (:233-277): a bilinear shrink
    of the full mel plus three crops whose start frames are drawn from `rng`.
    """

    def __init__(self, sample_rate: int = CLAP_SAMPLE_RATE, n_fft: int = CLAP_N_FFT,
                 hop_length: int = CLAP_HOP_LENGTH, n_mels: int = CLAP_N_MELS,
                 max_length_s: float = 10.0, rng: np.random.Generator | None = None):
        self.sample_rate = sample_rate
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.n_mels = n_mels
        self.max_length_s = max_length_s
        self.nb_max_samples = int(max_length_s * sample_rate)
        self.rng = rng if rng is not None else np.random.default_rng()

    def _mel(self, y2d: torch.Tensor) -> torch.Tensor:
        return clap_log_mel(y2d, self.sample_rate, self.n_fft, self.hop_length, self.n_mels)

    def _fit(self, clip: torch.Tensor) -> torch.Tensor:
        n = clip.shape[-1]
        if n < self.nb_max_samples:
            clip = clip.repeat(int(self.nb_max_samples / n))
        if clip.shape[-1] < self.nb_max_samples:
            clip = torch.nn.functional.pad(clip, (0, self.nb_max_samples - clip.shape[-1]))
        return clip[: self.nb_max_samples]

    def _fusion(self, clip: torch.Tensor) -> torch.Tensor:
        mel = self._mel(clip[None])[0]                          # [T, F]
        chunk = self.nb_max_samples // self.hop_length + 1
        total = mel.shape[0]
        if chunk >= total:
            return torch.stack([mel, mel, mel, mel], dim=0)
        ranges = np.array_split(list(range(0, total - chunk + 1)), 3)
        starts = [int(self.rng.choice(r)) if len(r) else 0 for r in ranges]
        crops = [mel[s:s + chunk] for s in starts]
        shrink = torch.nn.functional.interpolate(mel[None, None], size=[chunk, self.n_mels],
                                                 mode="bilinear", align_corners=False)[0, 0]
        return torch.stack([shrink, *crops], dim=0)

    def __call__(self, audio, sample_rate: int | None = None, return_tensors: str = "pt"):
        """audio [T], [B, T] or a list of 1-D clips -> dict(input_features
        [B, 4, frames, n_mels], is_longer [B, 1])."""
        if sample_rate is not None and sample_rate != self.sample_rate:
            raise NotImplementedError("resampling is outside this front-end; resample first")
        if isinstance(audio, (list, tuple)):
            clips = [as_device_f32(a, ndim_allowed=(1,)) for a in audio]
        else:
            a = as_device_f32(audio)
            clips = [a] if a.dim() == 1 else list(a)
        is_longer = [c.shape[-1] > self.nb_max_samples for c in clips]
        feats: list = [None] * len(clips)
        short = [i for i, l in enumerate(is_longer) if not l]
        if short:  # one batched launch for all short clips
            mel = self._mel(torch.stack([self._fit(clips[i]) for i in short]))
            for j, i in enumerate(short):
                feats[i] = mel[j][None].expand(4, -1, -1)
        for i, l in enumerate(is_longer):
            if l:
                feats[i] = self._fusion(clips[i])
        out = {"input_features": torch.stack(feats, dim=0),
               "is_longer": torch.tensor(is_longer, device=clips[0].device)[:, None]}
        if return_tensors == "np":
            out = {k: v.cpu().numpy() for k, v in out.items()}
        return out
