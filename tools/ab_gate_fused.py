import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, time
import importlib
sg = importlib.import_module("mlx_audio_amd.primitives.spectral_gate")
torch.manual_seed(0)
def run(x, **kw):
    sg._FUSE_INVERSE = True
    a = sg.spectral_gate(x, **kw)
    sg._FUSE_INVERSE = False
    b = sg.spectral_gate(x, **kw)
    sg._FUSE_INVERSE = True
    torch.cuda.synchronize()
    d = (a - b).abs().max().item()
    print(kw, tuple(x.shape), "max diff", d, "peak", b.abs().max().item(), flush=True)
    return d
t = torch.arange(22050 * 2, device="cuda") / 22050
base = 0.5 * torch.sin(2 * np.pi * 440 * t) * (t > 0.7)
for rows, n in [(1, 44100), (3, 44100), (5, 30000), (2, 1500), (1, 512), (7, 100000)]:
    x = torch.randn(rows, n, device="cuda") * 0.05
    x[:, :min(n, 44100)] += base[:min(n, 44100)]
    run(x)
    run(x, n_fft=1024)
    run(x, sr=16000)
    run(x, sr=44100, n_grad_freq=3, n_grad_time=5, prop_decrease=0.8)
    run(x, n_grad_freq=1, n_grad_time=1)
    run(x, freq_smooth_hz=0, time_smooth_ms=0)
    run(x, noise_profile=torch.randn(8000, device="cuda") * 0.05)
x = torch.randn(64, 220500, device="cuda") * 0.05
for fuse in (False, True, False, True):
    sg._FUSE_INVERSE = fuse
    for _ in range(3): sg.spectral_gate(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): sg.spectral_gate(x)
    torch.cuda.synchronize(); print("fuse", fuse, (time.perf_counter() - t0) / 20 * 1e3, "ms", flush=True)
