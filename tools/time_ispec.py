# Times htdemucs_ispec (model output planes -> audio) on 4 tracks x 4 sources x 60 s:
# python tools/time_ispec.py
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.frontends import htdemucs_ispec

B, S, C, F, Tf = 4, 4, 2, 2048, 2584
m = torch.randn((B, S, 2 * C, F, Tf), device="cuda")
length = 44100 * 60
for _ in range(3):
    y = htdemucs_ispec(m, length)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    y = htdemucs_ispec(m, length)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
print(f"htdemucs_ispec {tuple(m.shape)} -> {tuple(y.shape)}: {ms:.3f} ms ({m.numel() * 4 / 1e9:.2f} GB of planes)")
