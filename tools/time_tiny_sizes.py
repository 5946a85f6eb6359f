import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import istft, stft
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
y = 0.1 * torch.randn((16, 16000 * 30), device="cuda")
for n_fft in (16, 32, 64, 128, 256):
    hop = n_fft // 4
    S = stft(y, n_fft=n_fft, hop_length=hop)
    a = timed(lambda: stft(y, n_fft=n_fft, hop_length=hop))
    b = timed(lambda: istft(S, hop_length=hop, n_fft=n_fft, length=y.shape[1]))
    print(f"n_fft {n_fft:4d}: stft {a:8.3f} ms  istft {b:8.3f} ms   ({S.numel() * 8 / 1e9:.2f} GB of spectrum)")
