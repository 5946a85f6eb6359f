# htdemucs_normalize on the two layouts of the spectrogram: the strided view htdemucs_spec returns
# and a contiguous [B, C, F, T] array (frames fastest): python tools/time_normalize_layouts.py
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.frontends import htdemucs_normalize, htdemucs_spec
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
mix = 0.1 * torch.randn((16, 2, 44100 * 60), device="cuda")
spec = htdemucs_spec(mix)
specc = spec.contiguous()
print("strided view  ", timed(lambda: htdemucs_normalize(mix, spec)))
print("contiguous    ", timed(lambda: htdemucs_normalize(mix, specc)))
a = htdemucs_normalize(mix, spec)[0]; b = htdemucs_normalize(mix, specc)[0]
print("max diff", (a - b).abs().max().item())
