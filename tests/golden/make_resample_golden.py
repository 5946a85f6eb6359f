"""Writes tests/golden/resample_golden.npz: inputs and the outputs of the routines the
reference delegates to for sample-rate conversion (scipy.signal.resample_poly,
scipy.signal.resample, and its NumPy linear interpolation, reference
python/mlx_audio/primitives/resample.py:117-126,177-196,270-292), produced here with
SciPy 1.15.3 / NumPy 2.2 through oracle/resample.py.  Run from the repository root:
    python tests/golden/make_resample_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle.resample as orc  # noqa: E402

rng = np.random.default_rng(20240607)
t = np.arange(4410) / 44100.0
tone = (0.6 * np.sin(2 * np.pi * 440 * t) + 0.3 * np.sin(2 * np.pi * 3000 * t + 0.5)).astype(np.float32)
noise = rng.standard_normal(4410).astype(np.float32)
x = np.stack([tone, noise])
out = {"x": x}
for up, down in [(160, 441), (1, 2), (3, 1), (147, 160)]:
    out[f"poly_{up}_{down}"] = orc.resample_poly(x, up, down)
out["poly_160_441_reflect"] = orc.resample_poly(x, 160, 441, padtype="reflect")
for orig, target in [(44100, 16000), (16000, 44100), (44100, 22050)]:
    out[f"linear_{orig}_{target}"] = orc.resample(x, orig, target, res_type="linear")
    out[f"fft_{orig}_{target}"] = orc.resample(x, orig, target, res_type="fft")
out["fft_44100_16000_scaled_nofix"] = orc.resample(x[:, :4409], 44100, 16000, res_type="fft", fix=False, scale=True)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resample_golden.npz")
np.savez_compressed(path, **out)
print(path, {k: v.shape for k, v in out.items()})
