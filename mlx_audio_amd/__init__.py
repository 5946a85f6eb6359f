"""MI355X-native spectral front-end: a drop-in for the `mlx_audio.primitives`
stft / istft / melspectrogram / mel_filterbank / mfcc path and its model-side
callers (Whisper log-mel, HTDemucs spec/ispec, CLAP features).

Device arrays are PyTorch-ROCm tensors; the arithmetic runs in hand-written
HIP kernels (csrc/) reached through the C ABI of include/mia.h.
"""

from ._lib import HAS_HIP_EXT, has_hip_extensions  # noqa: F401

__version__ = "0.1.0"
