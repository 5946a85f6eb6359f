"""Spectrogram statistics over ~2 GB: ms per call (centroid, bandwidth, flatness, all four in one launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import (spectral_bandwidth, spectral_centroid, spectral_flatness,
                                      spectrogram_statistics)
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for bins, F in ((1025, 938), (2049, 500), (201, 3001)):
    B = max(1, int(2e9 / (bins * F * 4)))
    S = torch.rand((B, bins, F), device="cuda") + 0.05
    kw = dict(S=S, sr=22050, n_fft=2 * (bins - 1))
    c = spectral_centroid(**kw)
    row = [f"{B:5d} x {bins:5d} x {F:5d}"]
    for name, fn in (("centroid", lambda: spectral_centroid(**kw)),
                     ("bandwidth", lambda: spectral_bandwidth(centroid=c, **kw)),
                     ("flatness", lambda: spectral_flatness(S=S)),
                     ("flatness p=1", lambda: spectral_flatness(S=S, power=1.0)),
                     ("all four", lambda: spectrogram_statistics(S, sr=22050, n_fft=2 * (bins - 1)))):
        row.append(f"{name} {timed(fn):.3f}")
    print("  ".join(row) + "  ms")
    del S
