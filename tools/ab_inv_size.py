"""Time and check one inverse plan: python tools/ab_inv_size.py N_FFT   (hop = n_fft / 4; library = the one in place)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import istft, stft

n_fft = int(sys.argv[1])
hop = n_fft // 4
g = torch.Generator(device="cuda").manual_seed(1)
y = 0.1 * torch.randn((256, 16000 * 30), device="cuda", generator=g)
S = stft(y, n_fft=n_fft, hop_length=hop, pad_mode="reflect")


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


x = istft(S, hop_length=hop, length=y.shape[1])
err = float((x - y).abs().max())
print(f"n_fft {n_fft}: istft {timed(lambda: istft(S, hop_length=hop, length=y.shape[1])):.3f} ms  round-trip max err {err:.2e}")
