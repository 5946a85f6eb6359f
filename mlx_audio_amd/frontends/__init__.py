"""Model-side feature functions built on the primitives: Whisper log-mel,
HTDemucs spec / ispec, CLAP log-mel features."""

from .clap import CLAPFeatureExtractor, clap_log_mel
from .demucs import htdemucs_ispec, htdemucs_normalize, htdemucs_spec
from .whisper import (WhisperEncoderStem, compute_log_mel_spectrogram, pad_or_trim, sinusoids,
                      whisper_encoder_stem)

__all__ = [
    "compute_log_mel_spectrogram", "pad_or_trim", "htdemucs_spec", "htdemucs_ispec", "htdemucs_normalize",
    "clap_log_mel", "CLAPFeatureExtractor", "WhisperEncoderStem", "whisper_encoder_stem", "sinusoids",
]
