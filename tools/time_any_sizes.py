"""mel / stft at n_fft values without a compiled plan (whole-row FFT engine), 64 x 30 s, hop n_fft / 4: ms per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import melspectrogram, stft
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
y = 0.1 * torch.randn((64, 16000 * 30), device="cuda")
for n_fft in (320, 441, 480, 640, 882, 960, 1000, 1200, 1280, 1536, 1600, 1764, 2000, 1024):
    hop = n_fft // 4
    a = timed(lambda: melspectrogram(y, sr=16000, n_fft=n_fft, hop_length=hop, n_mels=80))
    b = timed(lambda: stft(y, n_fft=n_fft, hop_length=hop))
    print(f"n_fft {n_fft:5d}: mel {a:8.3f} ms  stft {b:8.3f} ms")
