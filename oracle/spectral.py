"""CPU oracle for the spectral front-end (TEST INFRASTRUCTURE, not product code).

A NumPy restatement of the reference's `mlx_audio.primitives` algorithms
(stft / istft / mel_filterbank / melspectrogram / power_to_db / dct / mfcc and
the window builders).  Every function cites the reference file:line it follows;
paths are relative to the reference checkout's `python/mlx_audio/`.

PARITY UNPINNED: the reference cannot be imported in the build container (its
array library, MLX, is absent) and its tests hold tolerances and recipes but no
stored numbers for this path.  The oracle is therefore pinned only by
  * independent implementations available offline (CPU torch.stft/istft,
    scipy.signal.get_window, scipy.fft.dct, transformers.audio_utils),
  * the whole Whisper and CLAP feature chains as Hugging Face transformers
    computes them (the code behind WhisperFeatureExtractor / ClapFeatureExtractor;
    tests/golden/hf_features_golden.npz, tests/test_golden.py),
  * the analytic known-answer cases the reference's own tests use, and
  * the reference's published tolerances,
all exercised in tests/test_oracle_*.py and tests/test_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product path (mlx_audio_amd) never does.

Precision model.  `dtype=np.float32` reproduces the reference's arithmetic
types: tables are built in float64 and cast, everything after is float32 /
complex64 (NumPy >= 2 keeps float32 through numpy.fft, which is pocketfft).
`dtype=np.float64` gives the "truth" used for error budgets.
"""

from __future__ import annotations

import math

import numpy as np

# constants/spectral.py:31-53, constants/algorithms.py:28-47
SLANEY_F_MIN = 0.0
SLANEY_F_SP = 200.0 / 3
SLANEY_MIN_LOG_HZ = 1000.0
SLANEY_LOGSTEP = math.log(6.4) / 27.0
SLANEY_MIN_LOG_MEL = (SLANEY_MIN_LOG_HZ - SLANEY_F_MIN) / SLANEY_F_SP
HTK_MEL_FACTOR = 2595.0
HTK_MEL_BASE = 700.0
FILTERBANK_EPSILON = 1e-10
WINDOW_SUM_EPSILON = 1e-8  # primitives/stft.py:21


# ---------------------------------------------------------------------------
# Windows — primitives/windows.py
# ---------------------------------------------------------------------------

_COSINE_COEFFS = {  # windows.py:63-67
    "hann": (0.5, 0.5),
    "hamming": (0.54, 0.46),
    "blackman": (0.42, 0.5, 0.08),
}
_WINDOW_ALIASES = {  # windows.py:112-122
    "hann": "hann",
    "hanning": "hann",
    "hamming": "hamming",
    "blackman": "blackman",
    "bartlett": "bartlett",
    "triangular": "bartlett",
    "rectangular": "rectangular",
    "boxcar": "rectangular",
    "ones": "rectangular",
}


def _symmetric_window(kind: str, n: int) -> np.ndarray:
    """Symmetric window of n points in float64, cast to float32.

    Follows windows.py:19-58 (generalised cosine sum with n-1 denominator,
    Blackman clamped at 0), :92-103 (Bartlett) and :106-108 (rectangular).
    """
    if kind == "rectangular" or n <= 1:
        return np.ones(n, dtype=np.float32)
    k = np.arange(n, dtype=np.float64)
    if kind == "bartlett":
        return (1 - np.abs(2 * k / (n - 1) - 1)).astype(np.float32)
    coeffs = _COSINE_COEFFS[kind]
    w = np.full(n, coeffs[0], dtype=np.float64)
    for i, c in enumerate(coeffs[1:], 1):
        sign = -1 if i % 2 == 1 else 1
        w = w + sign * c * np.cos(2 * i * np.pi * k / (n - 1))
    if kind == "blackman":
        w = np.maximum(w, 0.0)
    return w.astype(np.float32)


def get_window(window, n_fft: int, fftbins: bool = True) -> np.ndarray:
    """windows.py:189-253 (+ the NumPy fallback :166-186).

    Periodic (fftbins=True) = symmetric window of n_fft+1 points without its
    last sample.  Array windows must have length n_fft.
    """
    if isinstance(window, np.ndarray):
        if window.shape[0] != n_fft:
            raise ValueError(
                f"Window array length ({window.shape[0]}) must match n_fft ({n_fft})"
            )
        return window.astype(np.float32)
    if not isinstance(window, str):
        raise TypeError(f"window must be str or mx.array, got {type(window).__name__}")
    kind = _WINDOW_ALIASES.get(window.lower())
    if kind is None:
        supported = sorted(set(_WINDOW_ALIASES.keys()))
        raise ValueError(
            f"Unknown window type: '{window.lower()}'. Supported: {', '.join(supported)}"
        )
    n = n_fft + 1 if fftbins else n_fft
    return _symmetric_window(kind, n)[:n_fft]


def padded_window(window, win_length: int, n_fft: int) -> np.ndarray:
    """stft.py:88-106: window of win_length, zero-padded centred to n_fft."""
    win = get_window(window, win_length, fftbins=True)
    if win_length < n_fft:
        left = (n_fft - win_length) // 2
        win = np.pad(win, (left, n_fft - win_length - left))
    return win


# ---------------------------------------------------------------------------
# Pad / frame — primitives/stft.py:434-468, primitives/_frame_impl.py:79-125
# ---------------------------------------------------------------------------


def pad_signal(y: np.ndarray, pad_length: int, mode: str) -> np.ndarray:
    """stft.py:434-468 for a (batch, samples) array."""
    if mode == "constant":
        return np.pad(y, [(0, 0), (pad_length, pad_length)], mode="constant")
    if mode == "edge":
        return np.pad(y, [(0, 0), (pad_length, pad_length)], mode="edge")
    if mode == "reflect":
        if pad_length > 0:
            left = y[:, pad_length:0:-1]
            right = y[:, -2 : -pad_length - 2 : -1]
        else:
            left = y[:, :0]
            right = y[:, :0]
        return np.concatenate([left, y, right], axis=1)
    raise ValueError(f"Unknown pad_mode: '{mode}'. Supported: reflect, constant, edge")


def frame_signal(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    """_frame_impl.py:79-125: (batch, T) -> (batch, n_frames, frame_length)."""
    _, signal_length = y.shape
    if frame_length <= 0:
        raise ValueError(f"frame_length must be positive, got {frame_length}")
    if hop_length <= 0:
        raise ValueError(f"hop_length must be positive, got {hop_length}")
    if signal_length < frame_length:
        raise ValueError(
            f"Signal length ({signal_length}) must be >= frame_length "
            f"({frame_length}). Consider padding the signal."
        )
    n_frames = 1 + (signal_length - frame_length) // hop_length
    view = np.lib.stride_tricks.sliding_window_view(y, frame_length, axis=1)
    return view[:, ::hop_length][:, :n_frames]


# ---------------------------------------------------------------------------
# STFT / iSTFT — primitives/stft.py
# ---------------------------------------------------------------------------


def stft(
    y,
    n_fft: int = 2048,
    hop_length=None,
    win_length=None,
    window="hann",
    center: bool = True,
    pad_mode: str = "constant",
    dtype=np.float32,
) -> np.ndarray:
    """stft.py:136-222 with the compiled core :118-130.

    pad(n_fft//2) -> frame -> frames * win -> rfft(axis=-1) -> transpose to
    (batch, bins, frames); 1-D input loses the batch axis.
    """
    if hop_length is None:
        hop_length = n_fft // 4
    if win_length is None:
        win_length = n_fft
    if hop_length <= 0:
        raise ValueError(f"hop_length must be positive, got {hop_length}")
    if win_length <= 0:
        raise ValueError(f"win_length must be positive, got {win_length}")
    if win_length > n_fft:
        raise ValueError(f"win_length ({win_length}) must be <= n_fft ({n_fft})")
    if hop_length > n_fft:
        raise ValueError(
            f"hop_length ({hop_length}) should typically be <= n_fft ({n_fft})"
        )
    y = np.asarray(y, dtype=dtype)
    is_1d = y.ndim == 1
    if is_1d:
        y = y[None, :]
    win = padded_window(window, win_length, n_fft).astype(dtype)
    if center:
        y = pad_signal(y, n_fft // 2, pad_mode)
    frames = frame_signal(y, n_fft, hop_length) * win
    spec = np.fft.rfft(frames, axis=-1)
    spec = np.transpose(spec, (0, 2, 1))
    return spec[0] if is_1d else spec


def overlap_add(
    frames: np.ndarray, window: np.ndarray, hop_length: int, output_length: int
) -> np.ndarray:
    """stft.py:548-596 / csrc/metal/overlap_add.metal:27-54.

    out[b,i] = sum_f w[i-f*hop]*frames[b,f,i-f*hop] / max(sum_f w[i-f*hop]^2, 1e-8)
    accumulated over contributing frames in ascending f, in the array dtype.
    """
    batch, n_frames, n_fft = frames.shape
    dt = frames.dtype
    w = window.astype(dt)
    w2 = w * w
    acc = np.zeros((batch, output_length), dtype=dt)
    wsum = np.zeros(output_length, dtype=dt)
    for f in range(n_frames):
        start = f * hop_length
        if start >= output_length:
            break
        n = min(n_fft, output_length - start)
        acc[:, start : start + n] += w[:n] * frames[:, f, :n]
        wsum[start : start + n] += w2[:n]
    return acc / np.maximum(wsum, dt.type(WINDOW_SUM_EPSILON))


def istft(
    stft_matrix,
    hop_length=None,
    win_length=None,
    n_fft=None,
    window="hann",
    center: bool = True,
    length=None,
    dtype=np.float32,
) -> np.ndarray:
    """stft.py:225-344."""
    cdtype = np.complex64 if dtype == np.float32 else np.complex128
    stft_matrix = np.asarray(stft_matrix, dtype=cdtype)
    if stft_matrix.ndim not in (2, 3):
        raise ValueError(f"stft_matrix must be 2D or 3D, got {stft_matrix.ndim}D")
    is_2d = stft_matrix.ndim == 2
    if is_2d:
        stft_matrix = stft_matrix[None, :]
    _, freq_bins, n_frames = stft_matrix.shape
    if n_fft is None:
        n_fft = 2 * (freq_bins - 1)
    if hop_length is None:
        hop_length = n_fft // 4
    if win_length is None:
        win_length = n_fft
    win = padded_window(window, win_length, n_fft).astype(dtype)
    frames = np.fft.irfft(np.transpose(stft_matrix, (0, 2, 1)), n=n_fft, axis=-1)
    frames = frames.astype(dtype, copy=False)
    if length is not None:
        padded_length = length + n_fft if center else length
    else:
        padded_length = n_fft + (n_frames - 1) * hop_length
    y = overlap_add(frames, win, hop_length, padded_length)
    if center:
        pad = n_fft // 2
        if length is not None:
            y = y[:, pad : pad + length]
        else:
            end = y.shape[1] - pad
            y = y[:, pad:end] if end > pad else y[:, :0]
    elif length is not None:
        cur = y.shape[1]
        if length < cur:
            y = y[:, :length]
        elif length > cur:
            y = np.pad(y, [(0, 0), (0, length - cur)])
    return y[0] if is_2d else y


def magnitude(stft_matrix: np.ndarray) -> np.ndarray:
    """stft.py:347-362."""
    return np.abs(stft_matrix)


def phase(stft_matrix: np.ndarray) -> np.ndarray:
    """stft.py:365-379."""
    return np.arctan2(stft_matrix.imag, stft_matrix.real)


def check_nola(window, hop_length: int, n_fft: int, tol: float = 1e-10) -> bool:
    """stft.py:382-431 (scipy-style bin sums of the squared window)."""
    win = get_window(window, n_fft, fftbins=True)
    step = hop_length
    n_bins = n_fft // step
    binsums = sum(win[i * step : (i + 1) * step] ** 2 for i in range(n_bins))
    if n_fft % step != 0:
        binsums[: n_fft % step] += win[-(n_fft % step) :] ** 2
    return bool(np.min(binsums) > tol)


# ---------------------------------------------------------------------------
# Mel — primitives/mel.py
# ---------------------------------------------------------------------------


def hz_to_mel(frequencies, htk: bool = False) -> np.ndarray:
    """mel.py:42-73."""
    frequencies = np.asarray(frequencies)
    if htk:
        return HTK_MEL_FACTOR * np.log10(1.0 + frequencies / HTK_MEL_BASE)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(
            frequencies < SLANEY_MIN_LOG_HZ,
            (frequencies - SLANEY_F_MIN) / SLANEY_F_SP,
            SLANEY_MIN_LOG_MEL + np.log(frequencies / SLANEY_MIN_LOG_HZ) / SLANEY_LOGSTEP,
        )


def mel_to_hz(mels, htk: bool = False) -> np.ndarray:
    """mel.py:76-104."""
    mels = np.asarray(mels)
    if htk:
        return HTK_MEL_BASE * (10.0 ** (mels / HTK_MEL_FACTOR) - 1.0)
    return np.where(
        mels < SLANEY_MIN_LOG_MEL,
        SLANEY_F_MIN + SLANEY_F_SP * mels,
        SLANEY_MIN_LOG_HZ * np.exp(SLANEY_LOGSTEP * (mels - SLANEY_MIN_LOG_MEL)),
    )


def mel_filterbank(
    sr: int,
    n_fft: int,
    n_mels: int = 128,
    fmin: float = 0.0,
    fmax=None,
    htk: bool = False,
    norm="slaney",
) -> np.ndarray:
    """mel.py:182-253 (validation) and :111-179 (the NumPy builder).

    Note the order: eps-regularised slopes in float64 -> cast to float32 ->
    slaney scale applied in place on the float32 array (:168-174).
    """
    if n_mels <= 0:
        raise ValueError(f"n_mels must be positive, got {n_mels}")
    if fmin < 0:
        raise ValueError(f"fmin must be non-negative, got {fmin}")
    if fmax is None:
        fmax = sr / 2.0
    if fmin >= fmax:
        raise ValueError(f"fmin ({fmin}) must be less than fmax ({fmax})")
    if fmax > sr / 2.0:
        raise ValueError(f"fmax ({fmax}) cannot exceed Nyquist frequency ({sr / 2.0})")
    n_freqs = 1 + n_fft // 2
    fft_freqs = np.linspace(0, sr / 2.0, n_freqs)
    mel_points = np.linspace(hz_to_mel(fmin, htk=htk), hz_to_mel(fmax, htk=htk), n_mels + 2)
    hz_points = mel_to_hz(mel_points, htk=htk)
    f_lower = hz_points[:-2, None]
    f_center = hz_points[1:-1, None]
    f_upper = hz_points[2:, None]
    freqs = fft_freqs[None, :]
    lower = (freqs - f_lower) / (f_center - f_lower + FILTERBANK_EPSILON)
    upper = (f_upper - freqs) / (f_upper - f_center + FILTERBANK_EPSILON)
    fb = np.maximum(0, np.minimum(lower, upper)).astype(np.float32)
    if norm == "slaney":
        enorm = 2.0 / (hz_points[2 : n_mels + 2] - hz_points[:n_mels])
        fb *= enorm[:, None]
    elif norm is not None:
        raise ValueError(f"Unknown norm: '{norm}'. Supported: 'slaney', None")
    return fb


def melspectrogram(
    y,
    sr: int = 22050,
    n_fft: int = 2048,
    hop_length=None,
    win_length=None,
    window="hann",
    center: bool = True,
    pad_mode: str = "constant",
    power: float = 2.0,
    n_mels: int = 128,
    fmin: float = 0.0,
    fmax=None,
    htk: bool = False,
    norm="slaney",
    dtype=np.float32,
) -> np.ndarray:
    """mel.py:281-373 with the compiled apply step :265-276.

    stft -> abs -> power(., p) if p != 1 -> matmul(mel_basis, .)
    """
    S = stft(y, n_fft, hop_length, win_length, window, center, pad_mode, dtype=dtype)
    basis = mel_filterbank(sr, n_fft, n_mels, fmin, fmax, htk, norm).astype(dtype)
    mag = np.abs(S)
    if power != 1.0:
        mag = np.power(mag, dtype(power))
    return np.matmul(basis, mag)


# ---------------------------------------------------------------------------
# dB conversions — primitives/convert.py
# ---------------------------------------------------------------------------


def _to_db(S, ref, coefficient, amin, top_db):
    """convert.py:14-57.  The top_db clamp uses the max over the WHOLE array."""
    S = np.asarray(S)
    dt = S.dtype
    ref_value = ref(S) if callable(ref) else np.asarray(ref, dtype=dt)
    S = np.maximum(S, dt.type(amin))
    ref_value = np.maximum(ref_value, dt.type(amin))
    S_db = dt.type(coefficient) * np.log10(S / ref_value)
    if top_db is not None:
        if top_db <= 0:
            raise ValueError(f"top_db must be positive, got {top_db}")
        S_db = np.maximum(S_db, np.max(S_db) - dt.type(top_db))
    return S_db.astype(dt, copy=False)


def power_to_db(S, ref=1.0, amin: float = 1e-10, top_db=80.0):
    """convert.py:60-94."""
    return _to_db(S, ref, 10.0, amin, top_db)


def amplitude_to_db(S, ref=1.0, amin: float = 1e-5, top_db=80.0):
    """convert.py:129-163."""
    return _to_db(S, ref, 20.0, amin, top_db)


def db_to_power(S_db, ref: float = 1.0):
    """convert.py:97-126."""
    S_db = np.asarray(S_db)
    return (ref * np.power(S_db.dtype.type(10.0), S_db / S_db.dtype.type(10.0))).astype(S_db.dtype)


def db_to_amplitude(S_db, ref: float = 1.0):
    """convert.py:166-195."""
    S_db = np.asarray(S_db)
    return (ref * np.power(S_db.dtype.type(10.0), S_db / S_db.dtype.type(20.0))).astype(S_db.dtype)


# ---------------------------------------------------------------------------
# DCT / MFCC — primitives/mfcc.py
# ---------------------------------------------------------------------------


def dct_matrix(n_out: int, n_in: int, norm="ortho") -> np.ndarray:
    """mfcc.py:23-66: DCT-II basis built in float64, cast to float32."""
    n = np.arange(n_in)
    k = np.arange(n_out)
    basis = np.cos(np.pi * k[:, None] * (2 * n + 1) / (2 * n_in))
    if norm == "ortho":
        basis[0, :] *= 1.0 / np.sqrt(n_in)
        basis[1:, :] *= np.sqrt(2.0 / n_in)
    return basis.astype(np.float32)


def dct(x, type: int = 2, n=None, axis: int = -1, norm="ortho"):
    """mfcc.py:69-140 (Python fallback: moveaxis -> matmul(x, C.T) -> moveaxis)."""
    if type != 2:
        raise ValueError(f"Only DCT type 2 is supported, got {type}")
    x = np.asarray(x)
    input_size = x.shape[axis]
    if n is None:
        n = input_size
    C = dct_matrix(n, input_size, norm).astype(x.dtype)
    moved = axis != -1 and axis != x.ndim - 1
    if moved:
        x = np.moveaxis(x, axis, -1)
    out = np.matmul(x, C.T)
    if moved:
        out = np.moveaxis(out, -1, axis)
    return out


def mfcc(
    y=None,
    sr: int = 22050,
    S=None,
    n_mfcc: int = 20,
    dct_type: int = 2,
    norm="ortho",
    lifter: int = 0,
    n_fft: int = 2048,
    hop_length: int = 512,
    win_length=None,
    window="hann",
    center: bool = True,
    pad_mode: str = "constant",
    power: float = 2.0,
    n_mels: int = 128,
    fmin: float = 0.0,
    fmax=None,
    htk: bool = False,
    mel_norm="slaney",
    dtype=np.float32,
):
    """mfcc.py:143-281."""
    if n_mfcc <= 0:
        raise ValueError(f"n_mfcc must be positive, got {n_mfcc}")
    provided = S is not None
    if S is None:
        S = melspectrogram(
            y, sr, n_fft, hop_length, win_length, window, center, pad_mode,
            power, n_mels, fmin, fmax, htk, mel_norm, dtype=dtype,
        )
    S = np.asarray(S)
    batched = S.ndim == 3
    if not batched:
        S = S[None, :]
    S_db = S if provided else power_to_db(S, ref=1.0, amin=1e-10, top_db=80.0)
    M = dct(S_db, type=dct_type, n=n_mfcc, axis=1, norm=norm)
    if lifter > 0:
        n = np.arange(n_mfcc)
        lift = (1 + (lifter / 2.0) * np.sin(np.pi * (n + 1) / lifter)).astype(np.float32)
        M = M * lift.astype(M.dtype)[:, None]
    return M if batched else M[0]


# ---------------------------------------------------------------------------
# Model-side callers
# ---------------------------------------------------------------------------


def compute_log_mel_spectrogram(
    audio, n_mels: int = 80, n_fft: int = 400, hop_length: int = 160,
    sample_rate: int = 16000, dtype=np.float32,
):
    """models/whisper/inference.py:78-115.

    melspectrogram(fmin=0, fmax=8000) -> log10(max(., 1e-10)) ->
    max(., GLOBAL max - 8) -> (. + 4) / 4.  The max spans the whole array.
    """
    mel = melspectrogram(
        audio, sr=sample_rate, n_fft=n_fft, hop_length=hop_length, n_mels=n_mels,
        fmin=0.0, fmax=8000.0, dtype=dtype,
    )
    dt = mel.dtype.type
    log_mel = np.log10(np.maximum(mel, dt(1e-10)))
    log_mel = np.maximum(log_mel, np.max(log_mel) - dt(8.0))
    return (log_mel + dt(4.0)) / dt(4.0)


def pad_or_trim(audio: np.ndarray, length: int = 480000) -> np.ndarray:
    """models/whisper/inference.py:118-136."""
    if audio.shape[-1] > length:
        return audio[..., :length]
    if audio.shape[-1] < length:
        pad = [(0, 0)] * (audio.ndim - 1) + [(0, length - audio.shape[-1])]
        return np.pad(audio, pad)
    return audio


def htdemucs_spec(x: np.ndarray, n_fft: int = 4096, hop_length: int = 1024, dtype=np.float32):
    """models/demucs/model.py:378-417 (`HTDemucs._compute_stft`).

    x: [B, C, T] -> complex [B, C, n_fft//2, ceil(T/hop)].
    """
    B, C, T = x.shape
    le = int(math.ceil(T / hop_length))
    pad = hop_length // 2 * 3
    xp = np.pad(x, [(0, 0), (0, 0), (pad, pad + le * hop_length - T)])
    spec = stft(xp.reshape(B * C, -1), n_fft=n_fft, hop_length=hop_length,
                pad_mode="reflect", dtype=dtype)
    spec = spec / spec.real.dtype.type(math.sqrt(n_fft))
    spec = spec[:, :-1, :]
    spec = spec[:, :, 2 : 2 + le]
    return spec.reshape(B, C, spec.shape[1], spec.shape[2])


def htdemucs_ispec(m: np.ndarray, length: int, n_fft: int = 4096,
                   hop_length: int = 1024, dtype=np.float32):
    """models/demucs/model.py:432-506 (`HTDemucs._mask_and_istft`, CaC branch).

    m: real [B, S, 2C, F, Tf] laid out (re_ch0, im_ch0, re_ch1, ...) ->
    audio [B, S, C, length].
    """
    B, S, C2, F, Tf = m.shape
    C = C2 // 2
    mr = m.reshape(B, S, C, 2, F, Tf)
    z = mr[:, :, :, 0] + 1j * mr[:, :, :, 1]
    z = np.pad(z, [(0, 0), (0, 0), (0, 0), (0, 1), (0, 0)])
    z = np.pad(z, [(0, 0), (0, 0), (0, 0), (0, 0), (2, 2)])
    pad = hop_length // 2 * 3
    le = hop_length * int(math.ceil(length / hop_length)) + 2 * pad
    zf = z.reshape(B * S * C, z.shape[3], z.shape[4])
    x = istft(zf, hop_length=hop_length, length=le, dtype=dtype)
    x = x * x.dtype.type(math.sqrt(n_fft))
    x = x.reshape(B, S, C, -1)
    return x[:, :, :, pad : pad + length]


def clap_mel(audio: np.ndarray, sample_rate: int = 48000, n_fft: int = 1024,
             hop_length: int = 480, n_mels: int = 64, dtype=np.float32):
    """models/clap/feature_extractor.py:193-212 (`_compute_mel`) expressed with
    the primitives: librosa's melspectrogram defaults (hann, center, zero
    padding, power 2, slaney norm) equal the primitives' defaults
    (primitives/mel.py:281-296); power_to_db(ref=1, amin=1e-10, top_db=None);
    returned transposed as [T, F].
    """
    mel = melspectrogram(audio, sr=sample_rate, n_fft=n_fft, hop_length=hop_length,
                         n_mels=n_mels, fmin=50, fmax=14000, htk=True, dtype=dtype)
    db = power_to_db(mel, ref=1.0, amin=1e-10, top_db=None)
    return np.swapaxes(db, -1, -2)


def clap_padded_mel(audio: np.ndarray, sample_rate: int = 48000, n_fft: int = 1024,
                    hop_length: int = 480, n_mels: int = 64, max_length_s: float = 10.0,
                    dtype=np.float32):
    """models/clap/feature_extractor.py:214-231 + :176-177: tile a short clip up
    to nb_max_samples, zero-pad/truncate, one mel, stacked 4x -> [4, T, F]."""
    nb_max = int(max_length_s * sample_rate)
    if len(audio) < nb_max:
        audio = np.tile(audio, int(nb_max / len(audio)))
    if len(audio) < nb_max:
        audio = np.pad(audio, (0, nb_max - len(audio)), mode="constant")
    else:
        audio = audio[:nb_max]
    mel = clap_mel(audio, sample_rate, n_fft, hop_length, n_mels, dtype=dtype)
    return np.stack([mel, mel, mel, mel], axis=0)


def bilinear_resize(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """torch.nn.functional.interpolate(mode="bilinear", align_corners=False) on a 2-D float32
    array, which is what models/clap/feature_extractor.py:265-272 calls: source coordinate
    (dst + 0.5) * in/out - 0.5 clamped at 0, lower neighbour floor(), upper neighbour clamped to
    the last row / column, weights in float32, rows blended after columns."""
    img = np.asarray(img, np.float32)
    in_h, in_w = img.shape

    def axis(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        # scale * (dst + 0.5) - 0.5 with one rounding (the compiled library fuses it)
        src = (np.float64(scale) * (np.arange(n_out, dtype=np.float64) + 0.5) - 0.5).astype(np.float32)
        src = np.maximum(src, np.float32(0.0))
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, np.float32(1.0) - l1, l1

    h0, h1, hl0, hl1 = axis(in_h, out_h)
    w0, w1, wl0, wl1 = axis(in_w, out_w)
    top = img[h0][:, w0] * wl0 + img[h0][:, w1] * wl1
    bot = img[h1][:, w0] * wl0 + img[h1][:, w1] * wl1
    return (hl0[:, None] * top + hl1[:, None] * bot).astype(np.float32)


def clap_fusion_mel(audio: np.ndarray, rng, sample_rate: int = 48000, n_fft: int = 1024,
                    hop_length: int = 480, n_mels: int = 64, max_length_s: float = 10.0,
                    dtype=np.float32):
    """models/clap/feature_extractor.py:233-277 (`_compute_fusion_mel`): the full mel, three
    crops of chunk_frames frames whose starts are drawn from the thirds of the admissible
    range (front, middle, back, in that order; the reference draws from NumPy's global
    generator, here from `rng`), and the bilinear shrink of the full mel as channel 0.
    Returns ([4, chunk_frames, n_mels], (front, middle, back) start frames)."""
    mel = clap_mel(audio, sample_rate, n_fft, hop_length, n_mels, dtype=dtype)
    chunk = int(max_length_s * sample_rate) // hop_length + 1
    total = mel.shape[0]
    if chunk >= total:
        return np.stack([mel, mel, mel, mel], axis=0), (0, 0, 0)
    ranges = np.array_split(list(range(0, total - chunk + 1)), 3)
    if len(ranges[1]) == 0:
        ranges[1] = [0]
    if len(ranges[2]) == 0:
        ranges[2] = [0]
    starts = tuple(int(rng.choice(r)) for r in ranges)
    crops = [mel[s:s + chunk, :] for s in starts]
    shrink = bilinear_resize(mel.astype(np.float32), chunk, n_mels)
    return np.stack([shrink, *crops], axis=0).astype(np.float32), starts


def htdemucs_normalize(mix, spec, eps=1e-5):
    """models/demucs/model.py:206-233: complex-as-channels planes (re_ch0, im_ch0, re_ch1, ...),
    mean / population std over each batch item, std + eps; the waveform likewise."""
    mix = np.asarray(mix, np.float32)
    spec = np.asarray(spec, np.complex64)
    stacked = np.stack([spec.real, spec.imag], axis=2)
    B, C, _, F, T = stacked.shape
    mag = stacked.reshape(B, C * 2, F, T).astype(np.float32)
    spec_mean = np.mean(mag, axis=(1, 2, 3), keepdims=True)
    spec_std = np.std(mag, axis=(1, 2, 3), keepdims=True) + np.float32(eps)
    freq_in = (mag - spec_mean) / spec_std
    mix_mean = np.mean(mix, axis=(1, 2), keepdims=True)
    mix_std = np.std(mix, axis=(1, 2), keepdims=True) + np.float32(eps)
    return freq_in, spec_mean, spec_std, (mix - mix_mean) / mix_std, mix_mean, mix_std
