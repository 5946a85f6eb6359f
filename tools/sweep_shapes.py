"""Effective TB/s (bytes of the input spectrogram / time) of the spectrogram consumers over shapes, to spot
shapes a kernel's tiling does not like."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import (onset_strength, power_to_db, spectral_bandwidth, spectral_centroid, spectral_contrast,
                                      spectral_flatness, spectral_rolloff, dct, magnitude)
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for bins, F in ((1025, 938), (1025, 100), (1025, 10000), (201, 3001), (513, 1001), (2049, 500), (1025, 31), (129, 5000)):
    B = max(1, int(2e9 / (bins * F * 4)))
    S = torch.rand((B, bins, F), device="cuda") + 0.05
    gb = S.numel() * 4 / 1e9
    row = [f"{B:6d} x {bins:5d} x {F:6d}"]
    for name, fn in (("centroid", lambda: spectral_centroid(S=S, sr=22050, n_fft=2 * (bins - 1))),
                     ("rolloff", lambda: spectral_rolloff(S=S, sr=22050, n_fft=2 * (bins - 1))),
                     ("flatness", lambda: spectral_flatness(S=S)),
                     ("contrast", lambda: spectral_contrast(S=S, sr=22050, n_fft=2 * (bins - 1)) if bins >= 513 else None),
                     ("to_db", lambda: power_to_db(S))):
        try:
            t = timed(fn)
            row.append(f"{name} {gb / t:5.2f}")
        except Exception as e:
            row.append(f"{name} err")
    print("  ".join(row) + "  TB/s")
    del S
for n_mels, F in ((128, 938), (80, 3001), (40, 100), (64, 20000), (256, 500)):
    B = max(1, int(2e9 / (n_mels * F * 4)))
    S = torch.rand((B, n_mels, F), device="cuda") + 0.05
    gb = S.numel() * 4 / 1e9
    t1 = timed(lambda: onset_strength(S=S, sr=22050))
    t2 = timed(lambda: dct(S, n=20, axis=1))
    print(f"{B:6d} x {n_mels:4d} x {F:6d}  onset {gb / t1:5.2f}  dct->20 {gb / t2:5.2f}  TB/s")
    del S
