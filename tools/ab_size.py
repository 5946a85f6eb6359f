"""Time and check one compiled forward plan: python tools/ab_size.py N_FFT [HOP]   (library = the one in place)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import melspectrogram, stft

n_fft = int(sys.argv[1])
hop = int(sys.argv[2]) if len(sys.argv) > 2 else n_fft // 4
g = torch.Generator(device="cuda").manual_seed(1)
y = 0.1 * torch.randn((256, 16000 * 30), device="cuda", generator=g)


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


S = stft(y[:4], n_fft=n_fft, hop_length=hop, pad_mode="reflect")
ref = torch.stft(y[:4].double(), n_fft, hop_length=hop, window=torch.hann_window(n_fft, dtype=torch.float64, device="cuda"),
                 center=True, pad_mode="reflect", return_complex=True)
err = float((S - ref).abs().max() / ref.abs().max())
a = timed(lambda: melspectrogram(y, sr=16000, n_fft=n_fft, hop_length=hop, n_mels=128))
b = timed(lambda: stft(y, n_fft=n_fft, hop_length=hop))
print(f"n_fft {n_fft} hop {hop}: mel {a:.3f} ms  stft {b:.3f} ms  max err vs float64 {err:.2e}")
