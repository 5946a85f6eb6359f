"""Spectral gating noise reduction on MI355X with the signatures of
`mlx_audio.primitives.spectral_gate` (reference python/mlx_audio/primitives/spectral_gate.py).

The reference computes the STFT on its array library, moves magnitude and phase to the host,
builds the gate mask in NumPy / scipy.ndimage and goes back for the inverse STFT.  Here every
step stays on the GPU, the choice of the quietest tenth of the frames included (the order of a
stable sort of one number per frame), and the phase is never taken apart: magnitude * exp(i phase)
is the spectrum itself, so the gated spectrum is the STFT scaled by the mask.  The default call -
a per-bin threshold and at most two averages over bins and one over frames - is four launches:
the STFT with the frame energies beside it, the ranking of the frames, the threshold, and an
inverse STFT that builds and applies the mask as it reads the spectrum.
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._runtime import as_device_f32, get_plan, stream_ptr
from .stft import _resolve, istft, magnitude, stft
from .windows import padded_window_np


def _validate(n_fft, hop_length, prop_decrease) -> None:
    if n_fft <= 0:
        raise ValueError(f"n_fft must be positive, got {n_fft}")
    if hop_length <= 0:
        raise ValueError(f"hop_length must be positive, got {hop_length}")
    if prop_decrease < 0.0:
        raise ValueError(f"prop_decrease must be >= 0.0, got {prop_decrease}")
    if prop_decrease > 1.0:
        raise ValueError(f"prop_decrease must be <= 1.0, got {prop_decrease}")


_SPEC_MASK_MAX_WINDOW = 33   # bins the spectrum form of the mask kernel averages at most
_QUIET_MAX_FRAMES = 8192     # frames per row the ranking kernel takes
_FUSED_MAX_FRAMES, _FUSED_MAX_FRAMES_4096 = 7, 4   # time windows up to which the one-launch gate wins
_FUSE_INVERSE = True         # tests switch the one-launch gate + inverse STFT off to compare
_FUSE_ENERGY = True          # ... and the frame energies taken inside the forward kernel


class _Ops:
    """Thin launcher around the gating entry points for one device."""

    def __init__(self, device):
        self.dev = device
        self.lib = _lib.lib()

    def _run(self, name, *args):
        with torch.cuda.device(self.dev):
            _lib.check(getattr(self.lib, name)(*args, stream_ptr(self.dev)))

    def new(self, shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def frame_energy(self, mag):
        rows, bins, frames = mag.shape
        out = self.new((rows, frames))
        self._run("mia_frame_energy", mag.data_ptr(), rows, bins, frames, out.data_ptr())
        return out

    def noise_threshold(self, mag, idx, n_std):
        rows, bins, frames = mag.shape
        out = self.new((rows, bins))
        self._run("mia_noise_threshold", mag.data_ptr(), rows, bins, frames,
                  0 if idx is None else idx.data_ptr(), 0 if idx is None else idx.shape[1],
                  float(n_std), out.data_ptr())
        return out

    def time_min(self, x, size):
        rows, bins, frames = x.shape
        out = self.new(x.shape)
        self._run("mia_time_min_filter", x.data_ptr(), rows * bins, frames, int(size), out.data_ptr())
        return out

    def box(self, x, size_freq, size_time):
        """scipy.ndimage.uniform_filter(x, size=(1, size_freq, size_time), mode='nearest'):
        one moving average per axis whose size exceeds 1, frequency axis first."""
        rows, bins, frames = x.shape
        if size_freq > 1:
            out = self.new(x.shape)
            self._run("mia_box_filter", x.data_ptr(), rows, bins, frames, int(size_freq), out.data_ptr())
            x = out
        if size_time > 1:
            out = self.new(x.shape)
            self._run("mia_box_filter", x.data_ptr(), rows * bins, frames, 1, int(size_time), out.data_ptr())
            x = out
        return x

    def mask(self, mag, thr, gain):
        rows, bins, frames = mag.shape
        out = self.new(mag.shape)
        per_bin = thr.dim() == 2
        self._run("mia_gate_mask", mag.data_ptr(), thr.data_ptr(), rows, bins, frames, int(per_bin),
                  thr.shape[0], float(gain), out.data_ptr())
        return out

    def keep(self, mask, prop_decrease):
        out = self.new(mask.shape)
        self._run("mia_mask_op", mask.data_ptr(), 0, mask.numel(), 0, float(prop_decrease), out.data_ptr())
        return out

    def apply(self, mag, mask, ph, keep, prop_decrease, size_time=1):
        """mag * mask' * exp(i ph) as a (rows, bins, frames) view of a (rows, frames, bins) buffer;
        mask' = the mask averaged over size_time frames, then 1 - prop_decrease (1 - .) if keep."""
        rows, bins, frames = mag.shape
        out = torch.empty((rows, frames, bins), dtype=torch.complex64, device=self.dev)
        self._run("mia_gate_apply", mag.data_ptr(), mask.data_ptr(), ph.data_ptr(), rows, bins,
                  frames, int(bool(keep)), float(prop_decrease), int(size_time), out.data_ptr())
        return out.transpose(1, 2)

    def apply_spectrum(self, spec, mask, keep, prop_decrease, size_time=1, size_freq=1):
        """apply() from the complex (rows, bins, frames) spectrum the magnitudes were taken of
        (mag * exp(i ph) is that spectrum): spec * mask', same layout as apply()'s result; the
        mask may be averaged over size_freq bins before its time average, in the same launch."""
        rows, bins, frames = mask.shape
        rows_first = spec.transpose(1, 2)   # stft() hands out a view of (rows, frames, bins) rows
        if not rows_first.is_contiguous():
            rows_first = rows_first.contiguous()
        out = torch.empty((rows, frames, bins), dtype=torch.complex64, device=self.dev)
        self._run("mia_gate_apply_spectrum", rows_first.data_ptr(), mask.data_ptr(), rows, bins,
                  frames, int(bool(keep)), float(prop_decrease), int(size_freq), int(size_time),
                  out.data_ptr())
        return out.transpose(1, 2)

    def mask_box(self, mag, thr, gain, keep, prop_decrease, size_freq):
        """mask(), optionally keep(), and the moving average over size_freq bins in one launch."""
        rows, bins, frames = mag.shape
        out = self.new(mag.shape)
        self._run("mia_gate_mask_box", mag.data_ptr(), thr.data_ptr(), rows, bins, frames,
                  int(thr.dim() == 2), thr.shape[0], float(gain), int(bool(keep)),
                  float(prop_decrease), int(size_freq), out.data_ptr())
        return out

    # -- the same three steps straight from the complex (rows, frames, bins) spectrum ----------
    def frame_energy_spec(self, spec_rows):
        rows, frames, bins = spec_rows.shape
        out = self.new((rows, frames))
        self._run("mia_frame_energy_spec", spec_rows.data_ptr(), rows, bins, frames, out.data_ptr())
        return out

    def noise_threshold_spec(self, spec_rows, idx, n_std):
        rows, frames, bins = spec_rows.shape
        out = self.new((rows, bins))
        self._run("mia_noise_threshold_spec", spec_rows.data_ptr(), rows, bins, frames,
                  0 if idx is None else idx.data_ptr(), 0 if idx is None else idx.shape[1],
                  float(n_std), out.data_ptr())
        return out

    def mask_box_spec(self, spec_rows, thr, gain, keep, prop_decrease, size_freq):
        """mask_box() with the magnitudes taken from the spectrum; thr (rows or 1, bins)."""
        rows, frames, bins = spec_rows.shape
        out = self.new((rows, bins, frames))
        self._run("mia_gate_mask_box_spec", spec_rows.data_ptr(), thr.data_ptr(), rows, bins, frames,
                  thr.shape[0], float(gain), int(bool(keep)), float(prop_decrease), int(size_freq),
                  out.data_ptr())
        return out

    def gate(self, mag, spec, thr, gain, steps, keep_first, prop_decrease):
        """The mask pipeline with its first frequency average folded into the mask kernel and
        its last time average folded into the closing kernel, which scales the spectrum `spec`
        the magnitudes came from (mag None: the mask kernel takes them from `spec` itself).  steps: [(axis, size), ...] with axis "f" or "t", sizes > 1;
        keep_first: the keep step comes before the averages (the adaptive variant) instead of
        after them."""
        steps = list(steps)
        first = steps.pop(0)[1] if steps and steps[0][0] == "f" else 1
        last = steps.pop()[1] if steps and steps[-1][0] == "t" else 1
        if mag is None:   # no magnitude plane was made: the mask comes from the spectrum itself
            mask = self.mask_box_spec(spec.transpose(1, 2), thr, gain, keep_first, prop_decrease, first)
        else:
            mask = self.mask_box(mag, thr, gain, keep_first, prop_decrease, first)
        # a frequency average right before the closing kernel's time average rides along in it
        last_freq = steps.pop()[1] if steps and steps[-1][0] == "f" else 1
        for axis, size in steps:
            mask = self.box(mask, size, 1) if axis == "f" else self.box(mask, 1, size)
        return self.apply_spectrum(spec, mask, not keep_first, prop_decrease, last, last_freq)

    @staticmethod
    def fused_windows(steps):
        """(size_freq, size_freq2, size_time) when `steps` is at most an average over bins, a
        second one, and one over frames, in that order - the shape the inverse STFT kernel can
        apply by itself - else None."""
        steps = list(steps)
        first = steps.pop(0)[1] if steps and steps[0][0] == "f" else 1
        last = steps.pop()[1] if steps and steps[-1][0] == "t" else 1
        second = steps.pop()[1] if steps and steps[-1][0] == "f" else 1
        return None if steps else (first, second, last)

    def gated_istft(self, spec, thr, gain, windows, prop_decrease, n_fft, hop_length, win_length,
                    window, n_samples):
        """istft(gate(...), center=True, length=n_samples) in one launch, or None when the plan
        or the windows have no such kernel.  spec: the (rows, bins, frames) view stft() returns."""
        spec_rows = spec.transpose(1, 2)
        if not spec_rows.is_contiguous() or thr.dim() != 2:
            return None
        # the kernel keeps windows[2] rows of mask values in LDS: past these lengths it holds too few
        # workgroups per CU and the separate launches are the faster ones (tools/time_gate_inverse.py)
        if windows[2] > (_FUSED_MAX_FRAMES_4096 if n_fft >= 4096 else _FUSED_MAX_FRAMES):
            return None
        plan = get_plan(self.dev, n_fft, hop_length, True, "constant",
                        padded_window_np(window, win_length, n_fft))
        if not self.lib.mia_istft_gated_supported(plan.handle, *windows):
            return None
        rows, frames, _ = spec_rows.shape
        out = self.new((rows, n_samples + n_fft))
        self._run("mia_istft_gated", plan.handle, spec_rows.data_ptr(), rows, frames, out.shape[1],
                  1.0, thr.data_ptr(), thr.shape[0], float(gain), 1, float(prop_decrease),
                  int(windows[0]), int(windows[1]), int(windows[2]), out.data_ptr())
        return out[:, n_fft // 2:n_fft // 2 + n_samples]

    def quietest(self, energy, n_quiet):
        """Indices (rows, n_quiet) int32 of each row's n_quiet smallest energies, ascending, ties
        in frame order."""
        rows, frames = energy.shape
        if frames > _QUIET_MAX_FRAMES:   # long rows: a general (stable) sort
            return torch.sort(energy, dim=1, stable=True).indices[:, :n_quiet].to(torch.int32).contiguous()
        out = torch.empty((rows, n_quiet), dtype=torch.int32, device=self.dev)
        self._run("mia_quietest_frames", energy.data_ptr(), rows, frames, int(n_quiet), out.data_ptr())
        return out

    def stft_energy(self, x, n_fft, hop_length, win_length, window):
        """(stft(x, center=True) as its (rows, bins, frames) view, frame energies (rows, frames))
        from one launch, or None when the plan has no such kernel."""
        hop_length, win_length = _resolve(n_fft, hop_length, win_length)
        plan = get_plan(self.dev, n_fft, hop_length, True, "constant",
                        padded_window_np(window, win_length, n_fft))
        if not self.lib.mia_stft_energy_supported(plan.handle):
            return None
        rows, n = x.shape
        frames = plan.num_frames(n)
        spec = torch.empty((rows, frames, n_fft // 2 + 1), dtype=torch.complex64, device=self.dev)
        energy = self.new((rows, frames))
        self._run("mia_stft_energy", plan.handle, x.data_ptr(), rows, n, 1.0, spec.data_ptr(),
                  energy.data_ptr())
        return spec.transpose(1, 2), energy

    def times(self, a, b):
        out = self.new(a.shape)
        self._run("mia_mask_op", a.data_ptr(), b.data_ptr(), a.numel(), 1, 0.0, out.data_ptr())
        return out


def spectral_gate(audio, noise_profile=None, *, sr: int = 22050, n_fft: int = 2048,
                  hop_length=None, win_length=None, window: str = "hann",
                  threshold_db: float = -20.0, prop_decrease: float = 1.0,
                  n_std_thresh: float = 1.5, time_constant_s: float = 0.05,
                  freq_smooth_hz: float = 100.0, time_smooth_ms: float = 10.0,
                  stationary: bool = True, n_grad_freq: int = 2,
                  n_grad_time: int = 4) -> torch.Tensor:
    """Spectral gating noise reduction (reference spectral_gate.py:19-238)."""
    if hop_length is None:
        hop_length = n_fft // 4
    if win_length is None:
        win_length = n_fft
    _validate(n_fft, hop_length, prop_decrease)
    x = as_device_f32(audio)
    is_1d = x.dim() == 1
    if is_1d:
        x = x[None]
    n_samples = x.shape[-1]
    kw = dict(n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window, center=True)
    ops = _Ops(x.device)
    # the stationary estimate ranks the frames by energy: the forward kernel adds it up as it stores
    both = (ops.stft_energy(x.contiguous(), n_fft, hop_length, win_length, window)
            if noise_profile is None and stationary and _FUSE_ENERGY and x.shape[-1] > 0 else None)
    C, energy = both if both is not None else (stft(x, **kw), None)   # C: a (rows, bins, frames) view of (rows, frames, bins) rows
    rows, bins, frames = C.shape

    gain = 10 ** (threshold_db / 20.0)
    if noise_profile is None and not stationary:
        gain *= (1 + n_std_thresh)
    steps = []
    if freq_smooth_hz > 0 or time_smooth_ms > 0:
        fb = max(1, int(freq_smooth_hz * n_fft / sr))
        tf = max(1, int(time_smooth_ms * sr / (1000 * hop_length)))
        steps += [("f", fb), ("t", tf)]
    if n_grad_freq > 1 or n_grad_time > 1:
        steps += [("f", int(n_grad_freq)), ("t", int(n_grad_time))]
    steps = [(a, n) for a, n in steps if n > 1]

    # With a per-bin threshold (noise profile, or the stationary estimate) nothing needs the
    # magnitudes as an array: energy, threshold and mask read the spectrum's own rows.
    first_freq = steps[0][1] if steps and steps[0][0] == "f" else 1
    spec_rows = C.transpose(1, 2)
    direct = ((noise_profile is not None or stationary) and spec_rows.is_contiguous()
              and first_freq <= _SPEC_MASK_MAX_WINDOW)
    mag = None if direct else magnitude(C)

    if noise_profile is not None:
        noise = as_device_f32(noise_profile)
        noise_spec = stft(noise if noise.dim() > 1 else noise[None], **kw)
        if noise_spec.shape[0] not in (1, rows):
            raise ValueError(f"noise_profile has {noise_spec.shape[0]} rows, the audio {rows}")
        noise_rows = noise_spec.transpose(1, 2)
        # (the threshold straight from the noise spectrum's rows: the same bits as from its magnitude plane)
        thr = (ops.noise_threshold_spec(noise_rows, None, n_std_thresh) if noise_rows.is_contiguous()
               else ops.noise_threshold(magnitude(noise_spec), None, n_std_thresh))
    elif stationary:
        n_quiet = max(1, int(0.1 * frames))
        if energy is None:
            energy = ops.frame_energy_spec(spec_rows) if direct else ops.frame_energy(mag)
        quiet = ops.quietest(energy, n_quiet)
        thr = (ops.noise_threshold_spec(spec_rows, quiet, n_std_thresh) if direct
               else ops.noise_threshold(mag, quiet, n_std_thresh))
    else:
        tc_frames = max(1, int(time_constant_s * sr / hop_length))
        thr = ops.time_min(mag, tc_frames)

    y = None
    windows = ops.fused_windows(steps) if direct and _FUSE_INVERSE else None
    if windows is not None:   # mask, averages and inverse STFT in one launch
        y = ops.gated_istft(C, thr, gain, windows, prop_decrease, n_fft, hop_length, win_length,
                            window, n_samples)
    if y is None:
        enhanced = ops.gate(mag, C, thr, gain, steps, False, prop_decrease)
        y = istft(enhanced, hop_length=hop_length, win_length=win_length, window=window, center=True,
                  length=n_samples)
    return y[0] if is_1d else y


def spectral_gate_adaptive(audio, *, sr: int = 22050, n_fft: int = 2048, hop_length=None,
                           threshold_db: float = -20.0, prop_decrease: float = 1.0,
                           look_ahead_ms: float = 100.0,
                           look_back_ms: float = 100.0) -> torch.Tensor:
    """Spectral gating with a running-minimum noise floor (reference spectral_gate.py:241-353)."""
    if hop_length is None:
        hop_length = n_fft // 4
    _validate(n_fft, hop_length, prop_decrease)
    ahead = max(1, int(look_ahead_ms * sr / (1000 * hop_length)))
    back = max(1, int(look_back_ms * sr / (1000 * hop_length)))
    x = as_device_f32(audio)
    is_1d = x.dim() == 1
    if is_1d:
        x = x[None]
    n_samples = x.shape[-1]
    C = stft(x, n_fft=n_fft, hop_length=hop_length, center=True)
    mag = magnitude(C)
    ops = _Ops(mag.device)
    floor = ops.time_min(mag, ahead + back + 1)
    enhanced = ops.gate(mag, C, floor, 1.0 + 10 ** (threshold_db / 20.0), [("f", 3), ("t", 5)],
                        True, prop_decrease)
    y = istft(enhanced, hop_length=hop_length, center=True, length=n_samples)
    return y[0] if is_1d else y
