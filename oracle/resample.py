"""CPU oracle for sample-rate conversion (TEST INFRASTRUCTURE, not product code).

Restates `mlx_audio.primitives.resample` (reference
python/mlx_audio/primitives/resample.py).  That file does its arithmetic in
SciPy, a dependency of the reference that IS installed here (1.15.3, the
version this oracle was checked with):
  * `resample(res_type="fft")`      -> scipy.signal.resample        (resample.py:97-126)
  * `resample(res_type="linear")`   -> NumPy float64 interpolation  (resample.py:171-196)
  * `resample_poly`                 -> scipy.signal.resample_poly   (resample.py:270-292)
so the functions below call the same SciPy routines with the same arguments:
for this path the oracle is the reference's own arithmetic, PINNED by the
library the reference delegates to.  `upfirdn_f32` restates, operation by
operation, what that library does for float32 input (filter design, phase
layout, accumulation order); tests check it bit-for-bit against SciPy and use
it to state exactly what the GPU kernel has to reproduce.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product path (mlx_audio_amd) never does.
"""

from __future__ import annotations

import math

import numpy as np


def target_length(orig_length: int, orig_sr, target_sr, fix: bool = True) -> int:
    """resample.py:111-112."""
    ratio = target_sr / orig_sr
    return int(np.round(orig_length * ratio)) if fix else int(np.ceil(orig_length * ratio))


def resample(y, orig_sr, target_sr, res_type="fft", fix=True, scale=False, axis=-1):
    """resample.py:21-81 (validation), :84-136 (fft), :139-206 (linear)."""
    if orig_sr <= 0:
        raise ValueError(f"orig_sr must be positive, got {orig_sr}")
    if target_sr <= 0:
        raise ValueError(f"target_sr must be positive, got {target_sr}")
    y = np.asarray(y)
    if orig_sr == target_sr:
        return y
    if res_type not in ("fft", "linear"):
        raise ValueError(f"Unknown res_type: '{res_type}'. Supported: 'fft', 'linear'")
    if axis != -1:
        y = np.moveaxis(y, axis, -1)
    is_1d = y.ndim == 1
    if is_1d:
        y = y[None, :]
    _, n = y.shape
    ratio = target_sr / orig_sr
    m = target_length(n, orig_sr, target_sr, fix)
    if m == n:
        out = y
    elif res_type == "fft":
        from scipy.signal import resample as scipy_resample

        y_new = scipy_resample(y, m, axis=-1)
        if scale:
            y_new *= ratio
        out = y_new.astype(np.float32)
    else:
        idx = np.linspace(0, n - 1, m)
        lo = np.floor(idx).astype(np.int32)
        hi = np.minimum(lo + 1, n - 1)
        frac = idx - lo
        y_new = (1 - frac) * y[:, lo] + frac * y[:, hi]
        if scale:
            y_new *= ratio
        out = y_new.astype(np.float32)
    if is_1d:
        out = out[0]
    if axis != -1:
        out = np.moveaxis(out, -1, axis)
    return out


def resample_poly(y, up, down, axis=-1, padtype="constant"):
    """resample.py:209-302 (the SciPy branch; SciPy is a hard dependency here)."""
    if up <= 0:
        raise ValueError(f"up must be positive, got {up}")
    if down <= 0:
        raise ValueError(f"down must be positive, got {down}")
    g = math.gcd(up, down)
    up //= g
    down //= g
    y = np.asarray(y)
    if up == 1 and down == 1:
        return y
    if axis != -1:
        y = np.moveaxis(y, axis, -1)
    is_1d = y.ndim == 1
    if is_1d:
        y = y[None, :]
    from scipy.signal import resample_poly as scipy_resample_poly

    out = scipy_resample_poly(y, up, down, axis=-1, padtype=padtype).astype(np.float32)
    if is_1d:
        out = out[0]
    if axis != -1:
        out = np.moveaxis(out, -1, axis)
    return out


# ---------------------------------------------------------------------------
# What scipy.signal.resample_poly computes for float32 input, spelled out
# (scipy/signal/_signaltools.py resample_poly, _fir_filter_design.py firwin,
#  windows/_windows.py kaiser, _upfirdn.py + _upfirdn_apply.pyx).
# ---------------------------------------------------------------------------

def kaiser_lowpass(up: int, down: int) -> tuple[np.ndarray, int]:
    """The default anti-aliasing filter: firwin(20*max+1, 1/max, ('kaiser', 5.0)) in
    float64, cast to float32, times `up` in float32.  Returns (h, half_len)."""
    from scipy.special import i0

    max_rate = max(up, down)
    f_c = 1.0 / max_rate
    half_len = 10 * max_rate
    numtaps = 2 * half_len + 1
    alpha = 0.5 * (numtaps - 1)
    m = np.arange(0, numtaps) - alpha
    h = 0
    h += f_c * np.sinc(f_c * m)
    h -= 0.0 * np.sinc(0.0 * m)
    n = np.arange(0, numtaps)
    a = (numtaps - 1) / 2.0
    win = i0(5.0 * np.sqrt(1 - ((n - a) / a) ** 2.0)) / i0(5.0)
    h *= win
    c = np.cos(np.pi * m * 0.0)
    h /= np.sum(h * c)
    h = h.astype(np.float32)
    h *= up
    return h, half_len


def poly_layout(n_in: int, up: int, down: int, half_len: int, n_taps: int):
    """Zero padding of the filter and the slice of the upfirdn output that
    resample_poly keeps: (n_pre_pad, n_post_pad, n_pre_remove, n_out)."""
    def output_len(len_h):
        in_len_copy = n_in + (len_h + (-len_h % up)) // up - 1
        nt = in_len_copy * up
        return nt // down + (1 if nt % down > 0 else 0)

    n_out = n_in * up
    n_out = n_out // down + bool(n_out % down)
    n_pre_pad = down - half_len % down
    n_post_pad = 0
    n_pre_remove = (half_len + n_pre_pad) // down
    while output_len(n_taps + n_pre_pad + n_post_pad) < n_out + n_pre_remove:
        n_post_pad += 1
    return n_pre_pad, n_post_pad, n_pre_remove, n_out


def phase_table(h_padded: np.ndarray, up: int) -> np.ndarray:
    """(up, taps_per_phase): row t holds the taps of phase t, oldest input first."""
    padlen = len(h_padded) + (-len(h_padded) % up)
    full = np.zeros(padlen, h_padded.dtype)
    full[: len(h_padded)] = h_padded
    return np.ascontiguousarray(full.reshape(-1, up).T[:, ::-1])


def upfirdn_f32(x: np.ndarray, up: int, down: int) -> np.ndarray:
    """resample_poly(x, up, down) for one float32 row with constant (zero) padding:
    every output is a float32 sum, taken from the oldest input sample to the newest,
    of float32 products (a multiply, then an add: no fused multiply-add)."""
    x = np.asarray(x, np.float32)
    g = math.gcd(up, down)
    up //= g
    down //= g
    h, half_len = kaiser_lowpass(up, down)
    pre, post, n_pre_remove, n_out = poly_layout(len(x), up, down, half_len, len(h))
    hp = np.concatenate((np.zeros(pre, np.float32), h, np.zeros(post, np.float32)))
    table = phase_table(hp, up)
    per_phase = table.shape[1]
    out = np.zeros(n_out, np.float32)
    for o in range(n_out):
        q = (o + n_pre_remove) * down
        t, x_idx = q % up, q // up
        acc = np.float32(0.0)
        for j in range(per_phase):
            xi = x_idx - per_phase + 1 + j
            if 0 <= xi < len(x):
                acc = np.float32(acc + np.float32(x[xi] * table[t, j]))
        out[o] = acc
    return out
