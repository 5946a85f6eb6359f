"""Row-major DCT (128 bands -> n coefficients over (1024, 128, F)): ms per call by F and n."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import dct
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for F in (3001, 3008, 2944, 938):
    B = 1024 if F > 1000 else 4096
    S = torch.randn((B, 128, F), device="cuda")
    for n in (13, 20, 40):
        t = timed(lambda: dct(S, n=n, axis=1))
        gb = B * F * (128 + n) * 4 / 1e9
        print(f"F {F:5d} n {n:3d}: {t:.3f} ms  {gb / t:.2f} TB/s")
