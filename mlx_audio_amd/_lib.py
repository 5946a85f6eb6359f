"""ctypes loader for libmia_hip.so (the C ABI declared in include/mia.h).

Counterpart of the reference's extension loader
(python/mlx_audio/primitives/_extension.py:35-82): it exposes `HAS_HIP_EXT`,
`has_hip_extensions()` and the bound library.  Unlike the reference there is
no silent compute fallback: the host-side table builders work without a GPU,
but every device entry point raises if the library or a GPU is missing.
"""

from __future__ import annotations

import ctypes
import logging
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_logger = logging.getLogger(__name__)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmia_hip.so")

MIA_OK = 0
MIA_ERR_INVALID = -1
MIA_ERR_UNSUPPORTED = -2
MIA_ERR_HIP = -3

PAD_MODES = {"constant": 0, "edge": 1, "reflect": 2}
RS_PAD_MODES = {"constant": 0, "edge": 1, "reflect": 2, "symmetric": 3, "wrap": 4, "smooth": 5,
                "line": 6, "antisymmetric": 7, "antireflect": 8}
EPI_NONE, EPI_WHISPER, EPI_DB = 0, 1, 2
STEM_F16, STEM_F32 = 0, 1


class PlanDesc(ctypes.Structure):
    _fields_ = [
        ("n_fft", c_int),
        ("hop_length", c_int),
        ("center", c_int),
        ("pad_mode", c_int),
        ("window", POINTER(c_float)),
        ("n_mels", c_int),
        ("mel_basis", POINTER(c_float)),
        ("power", c_float),
        ("n_mfcc", c_int),
        ("dct_basis", POINTER(c_float)),
        ("lifter", POINTER(c_float)),
    ]


# name -> (restype, argtypes); must list every symbol include/mia.h declares.
SIGNATURES = {
    "mia_last_error": (c_char_p, []),
    "mia_version": (c_int, []),
    "mia_generate_window": (c_int, [c_char_p, c_int, c_int, POINTER(c_float)]),
    "mia_hz_to_mel": (c_int, [POINTER(c_double), c_int, c_int, POINTER(c_double)]),
    "mia_mel_to_hz": (c_int, [POINTER(c_double), c_int, c_int, POINTER(c_double)]),
    "mia_mel_filterbank": (c_int, [c_int, c_int, c_int, c_double, c_double, c_int, c_int,
                                   POINTER(c_float)]),
    "mia_dct_matrix": (c_int, [c_int, c_int, c_int, POINTER(c_float)]),
    "mia_plan_create": (c_int, [POINTER(PlanDesc), POINTER(c_void_p)]),
    "mia_plan_destroy": (c_int, [c_void_p]),
    "mia_plan_set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "mia_plan_get_info": (c_int, [c_void_p, c_char_p]),
    "mia_plan_num_frames": (c_int64, [c_void_p, c_int64]),
    "mia_fft_size_supported": (c_int, [c_int]),
    "mia_stft": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p, c_void_p]),
    "mia_stft_workspace_bytes": (c_size_t, [c_void_p, c_int64, c_int64]),
    "mia_stft_ws": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p, c_void_p,
                            c_size_t, c_void_p]),
    "mia_irfft_frames": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                                 c_void_p, c_void_p]),
    "mia_overlap_add": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int64,
                                c_float, c_void_p, c_void_p]),
    "mia_istft_workspace_bytes": (c_size_t, [c_void_p, c_int64, c_int64]),
    "mia_istft": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                          c_int64, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mia_mel_workspace_bytes": (c_size_t, [c_void_p, c_int64, c_int64]),
    "mia_melspectrogram": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_float, c_float,
                                   c_float, c_float, c_int, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "mia_melspectrogram_frames": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_float, c_float,
                                   c_float, c_float, c_int, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "mia_mel_frames_supported": (c_int, [c_void_p]),
    "mia_melspectrogram_static": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_float,
                                          c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "mia_mfcc_workspace_bytes": (c_size_t, [c_void_p, c_int64, c_int64]),
    "mia_mfcc": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_float, c_int, c_void_p,
                         c_void_p, c_size_t, c_void_p]),
    "mia_pad_signal": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mia_frame_signal": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "mia_to_db": (c_int, [c_void_p, c_int64, c_int64, c_float, c_float, c_float, c_float, c_int,
                          c_void_p, c_void_p, c_void_p]),
    "mia_from_db": (c_int, [c_void_p, c_int64, c_float, c_float, c_void_p, c_void_p]),
    "mia_dct2": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int, c_void_p,
                         c_void_p]),
    "mia_magnitude": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "mia_phase": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "mia_upfirdn": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int64,
                            c_int64, c_int, c_void_p, c_void_p]),
    "mia_upfirdn_margin": (c_int, [c_int, c_int, c_int, c_int64]),
    "mia_upfirdn_shared_lds_bytes": (c_int64, [c_int, c_int, c_int]),
    "mia_upfirdn_shared": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int,
                                   c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mia_resample_linear": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_double, c_double,
                                    c_void_p, c_void_p]),
    "mia_spectral_stat": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_int,
                                  c_float, c_int, c_float, c_float, c_float, c_void_p, c_void_p]),
    "mia_spectral_stats": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_float, c_int,
                                   c_float, c_float, c_float, c_void_p, c_void_p]),
    "mia_spectral_contrast": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int, c_int,
                                      c_int, c_void_p, c_void_p]),
    "mia_zero_crossing_rate": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                       c_int, c_void_p, c_void_p]),
    "mia_frame_energy": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "mia_noise_threshold": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int, c_float,
                                    c_void_p, c_void_p]),
    "mia_time_min_filter": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mia_box_filter": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mia_gate_mask": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_int64, c_float,
                              c_void_p, c_void_p]),
    "mia_gate_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int,
                               c_float, c_int, c_void_p, c_void_p]),
    "mia_gate_apply_spectrum": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int,
                                        c_float, c_int, c_int, c_void_p, c_void_p]),
    "mia_frame_energy_spec": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "mia_noise_threshold_spec": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int, c_float,
                                         c_void_p, c_void_p]),
    "mia_gate_mask_box_spec": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_float,
                                       c_int, c_float, c_int, c_void_p, c_void_p]),
    "mia_quietest_frames": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mia_stft_energy_supported": (c_int, [c_void_p]),
    "mia_stft_energy": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p, c_void_p,
                                c_void_p]),
    "mia_istft_gated_supported": (c_int, [c_void_p, c_int, c_int, c_int]),
    "mia_istft_gated": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float, c_void_p,
                                c_int64, c_float, c_int, c_float, c_int, c_int, c_int, c_void_p,
                                c_void_p]),
    "mia_gate_mask_box": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_int64,
                                  c_float, c_int, c_float, c_int, c_void_p, c_void_p]),
    "mia_mask_op": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p]),
    "mia_normalize_rows": (c_int, [c_void_p, c_int64, c_int64, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "mia_stft_moments": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p, c_int, c_int,
                                 c_int64, c_int64, c_void_p, c_void_p]),
    "mia_pad_rows_moments": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p,
                                     c_void_p]),
    "mia_normalize_rows_sums": (c_int, [c_void_p, c_int64, c_int64, c_float, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "mia_cac_to_spectrum": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "mia_normalize_cac_sums": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int64, c_int64, c_int64,
                                       c_int64, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mia_normalize_cac": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int64, c_int64, c_int64,
                                  c_int64, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mia_spectrum_part": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mia_transpose_spectrum": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "mia_griffinlim_init": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "mia_polar": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mia_griffinlim_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float,
                                      c_void_p]),
    "mia_onset_flux": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "mia_envelope_filter": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p,
                                    c_int64, c_void_p]),
    "mia_whisper_stem_weight_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mia_whisper_stem_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mia_whisper_stem_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int, c_int]),
    "mia_whisper_stem": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "mia_resample_fft_table_bytes": (c_size_t, [c_int64, c_int64]),
    "mia_resample_fft_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "mia_resample_fft_tables": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mia_resample_fft": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_float, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_void_p]),
}

HAS_HIP_EXT: bool = False
_lib = None
_load_error: str | None = None


def _load():
    global HAS_HIP_EXT, _lib, _load_error
    try:
        # One HIP runtime per process: PyTorch-ROCm bundles its own
        # libamdhip64.so.7 and the tensors and streams handed to the kernels
        # belong to it.  Importing torch first makes the dynamic loader resolve
        # libmia_hip.so's NEEDED libamdhip64.so.7 to that already-loaded copy;
        # the other order would put a second runtime (ROCm's) in the process.
        import torch  # noqa: F401

        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        # self-test, as the reference does with generate_window("hann", 4, True)
        buf = (c_float * 4)()
        if lib.mia_generate_window(b"hann", 4, 1, buf) != MIA_OK:
            raise OSError("self-test failed: " + lib.mia_last_error().decode())
        _lib = lib
        HAS_HIP_EXT = True
    except (OSError, AttributeError) as e:  # not built, or built without a symbol
        _lib = None
        HAS_HIP_EXT = False
        _load_error = str(e)
        _logger.info("libmia_hip.so not available: %s", e)


_load()


def has_hip_extensions() -> bool:
    """True if libmia_hip.so is loaded (mirrors `has_cpp_extensions`)."""
    return HAS_HIP_EXT


def lib():
    """The bound library; raises if it is not built (no CPU fallback exists)."""
    if _lib is None:
        raise RuntimeError(
            f"libmia_hip.so is not available ({_load_error}). Build it with "
            "`python mlx_audio_amd/csrc/build.py` (needs hipcc); the spectral "
            "front-end has no CPU fallback."
        )
    return _lib


def check(rc: int) -> None:
    """Map a C status to the exception type the reference raises."""
    if rc == MIA_OK:
        return
    msg = lib().mia_last_error().decode(errors="replace")
    if rc == MIA_ERR_INVALID:
        raise ValueError(msg)
    if rc == MIA_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)
