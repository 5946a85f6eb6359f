"""Sample-rate conversion on MI355X with the signatures of
`mlx_audio.primitives.resample` (reference python/mlx_audio/primitives/resample.py).

`resample_poly` and `resample(res_type="linear")` run as HIP kernels and return,
bit for bit, what the reference's host path (SciPy / NumPy on float32 rows) returns.
The anti-aliasing filter of `resample_poly` is designed here on the host, once per
(up, down), exactly as SciPy designs it (float64 windowed sinc, cast to float32).

`resample(res_type="fft")`, the default, transforms the whole row (any length) on the GPU
as chirp-z convolutions over power-of-two FFTs; it agrees with SciPy's float32 result to
float32 FFT round-off (a few 1e-6 of the row's peak), not bit for bit.
"""

from __future__ import annotations

import functools
import threading
from collections import OrderedDict
import math

import numpy as np
import torch

from .. import _lib
from .._runtime import require_gpu, stream_ptr

_BACKGROUND = ("mean", "median", "minimum", "maximum")


def _validate_positive(value, name: str) -> None:
    # primitives/_validation.py: validate_positive
    if value <= 0:
        raise ValueError(f"{name} must be positive, got {value}")


def _rows(y, axis: int):
    """Input as a contiguous float32 (rows, samples) device tensor + how to undo it."""
    require_gpu()
    if not isinstance(y, torch.Tensor):
        y = torch.as_tensor(np.asarray(y))
    if not y.is_cuda:
        y = y.to("cuda")
    if y.dtype != torch.float32:
        y = y.to(torch.float32)
    if axis != -1:
        y = torch.movedim(y, axis, -1)
    is_1d = y.dim() == 1
    if is_1d:
        y = y[None, :]
    if y.dim() != 2:  # the reference unpacks `batch_size, orig_length = y.shape`
        raise ValueError(f"expected a 1-D or 2-D signal, got {y.dim()}-D")
    return y.contiguous(), is_1d


def _restore(out: torch.Tensor, is_1d: bool, axis: int) -> torch.Tensor:
    if is_1d:
        out = out[0]
    if axis != -1:
        out = torch.movedim(out, -1, axis)
    return out


def resample(y, orig_sr: int, target_sr: int, res_type: str = "fft", fix: bool = True,
             scale: bool = False, axis: int = -1) -> torch.Tensor:
    """Resample audio to a different sample rate (reference resample.py:21-81).

    Both methods run on the GPU: "fft" (resample.py:84-136, SciPy's Fourier method over the
    whole signal) and "linear" (resample.py:139-206).
    """
    _validate_positive(orig_sr, "orig_sr")
    _validate_positive(target_sr, "target_sr")
    if orig_sr == target_sr:
        return y
    if res_type not in ("fft", "linear"):
        raise ValueError(f"Unknown res_type: '{res_type}'. Supported: 'fft', 'linear'")

    x, is_1d = _rows(y, axis)
    rows, n_in = x.shape
    ratio = target_sr / orig_sr
    n_out = int(np.round(n_in * ratio)) if fix else int(np.ceil(n_in * ratio))
    if n_out == n_in:
        return _restore(x, is_1d, axis)
    if n_in == 0:
        raise IndexError("cannot resample an empty signal")
    if res_type == "fft":
        if n_out < 1:  # scipy.signal.resample cannot produce an empty row either
            raise ValueError(f"resampling {n_in} samples by {ratio} leaves no output samples")
        return _restore(_resample_fft(x, n_out, float(ratio) if scale else 1.0), is_1d, axis)
    out = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
    # numpy.linspace(0, n_in - 1, n_out): i * ((n_in - 1) / (n_out - 1)), last point exact
    step = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mia_resample_linear(
            x.data_ptr(), rows, n_in, n_out, float(step), float(ratio) if scale else 1.0,
            out.data_ptr(), stream_ptr(x.device)))
    return _restore(out, is_1d, axis)


_FFT_WORKSPACE_BYTES = 8 << 30  # rows are processed in chunks that keep the workspace below this
# (n_in, n_out, device) -> (tables, event recorded behind the kernels that fill them).  The
# tables are built asynchronously on whichever stream was current at the first call; every
# later user makes ITS stream wait for that event, so a consumer on another stream never
# reads half-built tables.  Evicted entries stay alive as long as a caller holds them.
_fft_tables_cache: "OrderedDict[tuple, tuple]" = OrderedDict()
_fft_tables_lock = threading.Lock()
_FFT_TABLES_MAX = 8


def _fft_tables(lib, n_in: int, n_out: int, dev: torch.device, work: torch.Tensor):
    key = (n_in, n_out, dev.index)
    cur = torch.cuda.current_stream(dev)
    with _fft_tables_lock:
        hit = _fft_tables_cache.get(key)
        if hit is not None:
            _fft_tables_cache.move_to_end(key)
            tables, ready = hit
            cur.wait_event(ready)
            # The block belongs to the allocator pool of the stream that built it: tell the
            # allocator that this stream reads it too, or an evicted entry could be handed out
            # again while kernels queued here still read it.
            tables.record_stream(cur)
            return tables
        tables = torch.empty(lib.mia_resample_fft_table_bytes(n_in, n_out), dtype=torch.uint8,
                             device=dev)
        _lib.check(lib.mia_resample_fft_tables(n_in, n_out, tables.data_ptr(), work.data_ptr(),
                                               work.numel(), cur.cuda_stream))
        ready = torch.cuda.Event()
        ready.record(cur)
        _fft_tables_cache[key] = (tables, ready)
        while len(_fft_tables_cache) > _FFT_TABLES_MAX:
            _fft_tables_cache.popitem(last=False)
        return tables


def _resample_fft(x: torch.Tensor, n_out: int, gain: float) -> torch.Tensor:
    """scipy.signal.resample(x, n_out, axis=-1) * gain for float32 rows (resample.py:117-126)."""
    lib = _lib.lib()
    rows, n_in = x.shape
    dev = x.device
    per_row = lib.mia_resample_fft_workspace_bytes(1, n_in, n_out)
    if per_row == 0:
        raise NotImplementedError(
            f"resample(res_type='fft'): rows of {n_in} -> {n_out} samples exceed the supported 2^27")
    chunk = max(1, min(rows, _FFT_WORKSPACE_BYTES // per_row))
    out = torch.empty((rows, n_out), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = stream_ptr(dev)
        work = torch.empty(chunk * per_row, dtype=torch.uint8, device=dev)
        tables = _fft_tables(lib, n_in, n_out, dev, work)
        for r0 in range(0, rows, chunk):
            n = min(chunk, rows - r0)
            _lib.check(lib.mia_resample_fft(x[r0:r0 + n].data_ptr(), n, n_in, n_out, gain,
                                            tables.data_ptr(), out[r0:r0 + n].data_ptr(),
                                            work.data_ptr(), work.numel(), stream))
    return out


@functools.lru_cache(maxsize=64)
def _kaiser_lowpass(up: int, down: int):
    """SciPy's default resample_poly filter for float32 input:
    firwin(2*half_len+1, 1/max(up,down), window=('kaiser', 5.0)) in float64, cast to
    float32, then multiplied by `up` in float32.  Returns (h, half_len)."""
    from scipy.special import i0  # SciPy is a dependency of the reference for this function too

    max_rate = max(up, down)
    cutoff = 1.0 / max_rate
    half_len = 10 * max_rate
    numtaps = 2 * half_len + 1
    m = np.arange(0, numtaps) - 0.5 * (numtaps - 1)
    h = cutoff * np.sinc(cutoff * m)
    n = np.arange(0, numtaps)
    mid = (numtaps - 1) / 2.0
    h = h * (i0(5.0 * np.sqrt(1 - ((n - mid) / mid) ** 2.0)) / i0(5.0))
    h = h / np.sum(h)  # unit gain at DC
    h = h.astype(np.float32)
    h *= up
    h.setflags(write=False)
    return h, half_len


def _poly_layout(n_in: int, up: int, down: int, half_len: int, n_taps: int):
    """Zero padding in front of / behind the filter that centres the output on the input,
    the outputs of the full convolution to drop in front, and the output length."""
    def full_len(len_h: int) -> int:
        nt = (n_in + (len_h + (-len_h % up)) // up - 1) * up
        return nt // down + (1 if nt % down else 0)

    n_out = -(-(n_in * up) // down)
    pre = down - half_len % down
    post = 0
    skip = (half_len + pre) // down
    while full_len(n_taps + pre + post) < n_out + skip:
        post += 1
    return pre, post, skip, n_out


def _phase_table(h: np.ndarray, pre: int, post: int, up: int) -> np.ndarray:
    """(up, taps_per_phase) float32: row t = taps t, t+up, ... of the padded filter, reversed
    so that a row lines up with consecutive input samples, oldest first."""
    n = pre + len(h) + post
    full = np.zeros(n + (-n % up), np.float32)
    full[pre:pre + len(h)] = h
    return np.ascontiguousarray(full.reshape(-1, up).T[:, ::-1])


_table_dev_cache: dict = {}


def _phase_table_dev(up: int, down: int, pre: int, post: int, device, margin: int = 0) -> torch.Tensor:
    """The phase table on the device; with `margin`, every row between that many zeros."""
    key = (up, down, pre, post, device.index, margin)
    t = _table_dev_cache.get(key)
    if t is None:
        if len(_table_dev_cache) >= 64:
            _table_dev_cache.clear()
        h, _ = _kaiser_lowpass(up, down)
        table = _phase_table(h, pre, post, up)
        if margin:
            table = np.pad(table, ((0, 0), (margin, margin)))
        t = torch.from_numpy(np.ascontiguousarray(table)).to(device)
        _table_dev_cache[key] = t
    return t


def resample_poly(y, up: int, down: int, axis: int = -1, padtype: str = "constant") -> torch.Tensor:
    """Polyphase resampling by the integer ratio up/down (reference resample.py:209-302).

    padtype: any of SciPy's: "constant" (zeros), "edge", "reflect", "symmetric", "wrap", "line",
    "smooth", "antisymmetric", "antireflect" (the signal extended past its ends by that rule),
    or one of "mean", "median", "minimum", "maximum" (that statistic of each row is
    subtracted before the filter and added back after it).
    """
    _validate_positive(up, "up")
    _validate_positive(down, "down")
    g = math.gcd(up, down)
    up //= g
    down //= g
    if up == 1 and down == 1:
        return y
    if padtype not in _lib.RS_PAD_MODES and padtype not in _BACKGROUND:
        raise ValueError(
            "padtype must be one of: maximum, mean, median, minimum, constant, wrap, edge, "
            "smooth, symmetric, reflect, antisymmetric, antireflect, line")

    x, is_1d = _rows(y, axis)
    rows, n_in = x.shape
    if n_in == 0:
        raise ValueError("cannot resample an empty signal")
    h, half_len = _kaiser_lowpass(up, down)
    pre, post, skip, n_out = _poly_layout(n_in, up, down, half_len, len(h))
    taps_per_phase = -(-(pre + len(h) + post) // up)
    # 0: the per-output kernel; else the kernel that shares sample windows between adjacent
    # outputs, which wants its table rows between that many zeros
    margin = _lib.lib().mia_upfirdn_margin(up, down, taps_per_phase, n_out)
    table = _phase_table_dev(up, down, pre, post, x.device, margin)

    background = None
    mode = _lib.RS_PAD_MODES.get(padtype, 0)
    if padtype in _BACKGROUND:
        if padtype == "mean":
            background = x.mean(dim=-1, keepdim=True)
        elif padtype == "median":
            srt = x.sort(dim=-1).values  # mean of the two middle samples for even lengths
            background = ((srt[:, (n_in - 1) // 2] + srt[:, n_in // 2]) * 0.5)[:, None]
        elif padtype == "minimum":
            background = x.amin(dim=-1, keepdim=True)
        else:
            background = x.amax(dim=-1, keepdim=True)
        x = (x - background).contiguous()

    out = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        if margin:
            _lib.check(_lib.lib().mia_upfirdn_shared(
                x.data_ptr(), rows, n_in, table.data_ptr(), margin, up, down, taps_per_phase,
                skip, n_out, mode, out.data_ptr(), stream_ptr(x.device)))
        else:
            _lib.check(_lib.lib().mia_upfirdn(
                x.data_ptr(), rows, n_in, table.data_ptr(), up, down, taps_per_phase, skip,
                n_out, mode, out.data_ptr(), stream_ptr(x.device)))
    if background is not None:
        out += background
    return _restore(out, is_1d, axis)
