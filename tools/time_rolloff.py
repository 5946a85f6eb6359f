"""Roll-off frequency over ~2 GB spectrograms: ms per call and TB/s of input bytes, random columns
(every lane crosses near the same bin) and columns with one dominant peak at a random bin (every
lane crosses somewhere else: the widest re-read)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import spectral_rolloff
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for bins, F in ((1025, 938), (2049, 500), (513, 1001), (201, 3001), (1025, 31)):
    B = max(1, int(2e9 / (bins * F * 4)))
    S = torch.rand((B, bins, F), device="cuda") + 0.05
    gb = S.numel() * 4 / 1e9
    t = timed(lambda: spectral_rolloff(S=S, sr=22050, n_fft=2 * (bins - 1)))
    peak = torch.randint(0, bins, (B, 1, F), device="cuda")
    S.scatter_(1, peak, 1e4)
    t2 = timed(lambda: spectral_rolloff(S=S, sr=22050, n_fft=2 * (bins - 1)))
    print(f"{B:6d} x {bins:5d} x {F:6d}  random {t:.3f} ms {gb / t:5.2f} TB/s   scattered peaks {t2:.3f} ms {gb / t2:5.2f} TB/s")
    del S, peak
