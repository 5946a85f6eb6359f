"""Audio DSP primitives on MI355X with the signatures of `mlx_audio.primitives`
(reference python/mlx_audio/primitives/__init__.py:197-261, hot-path subset)."""

from .._lib import HAS_HIP_EXT, has_hip_extensions
from .convert import amplitude_to_db, db_to_amplitude, db_to_power, power_to_db
from .features import (fused_statistics, spectrogram_statistics, spectral_bandwidth, spectral_centroid, spectral_contrast,
                       spectral_flatness, spectral_rolloff, zero_crossing_rate)
from .griffinlim import griffinlim, griffinlim_iter
from .mel import hz_to_mel, mel_filterbank, mel_to_hz, melspectrogram
from .mfcc import dct, mfcc
from .onset import onset_detect, onset_strength, onset_strength_multi
from .resample import resample, resample_poly
from .spectral_gate import spectral_gate, spectral_gate_adaptive
from .stft import check_nola, istft, magnitude, phase, stft
from .windows import get_window

__all__ = [
    "HAS_HIP_EXT", "has_hip_extensions",
    "stft", "istft", "magnitude", "phase", "check_nola", "get_window",
    "mel_filterbank", "melspectrogram", "hz_to_mel", "mel_to_hz",
    "dct", "mfcc",
    "power_to_db", "amplitude_to_db", "db_to_power", "db_to_amplitude",
    "resample", "resample_poly",
    "spectral_centroid", "spectral_bandwidth", "spectral_rolloff", "spectral_flatness",
    "spectral_contrast", "zero_crossing_rate", "fused_statistics", "spectrogram_statistics",
    "onset_strength", "onset_detect", "onset_strength_multi",
    "griffinlim", "griffinlim_iter",
    "spectral_gate", "spectral_gate_adaptive",
]
