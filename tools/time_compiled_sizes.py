"""ms per call and frames per microsecond of the compiled forward plans on one shape of work each:
mel spectrogram (128 bands) of 256 x 30 s @16 kHz at hop = n_fft / 4 - to compare the plans with
each other (a plan that spills or has too few threads per frame shows as a low frame rate)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import melspectrogram, stft

y = 0.1 * torch.randn((256, 16000 * 30), device="cuda")


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


for n_fft in (256, 400, 512, 800, 1024, 2048, 4096):
    hop = n_fft // 4
    frames = 256 * (1 + y.shape[1] // hop)
    a = timed(lambda: melspectrogram(y, sr=16000, n_fft=n_fft, hop_length=hop, n_mels=128))
    b = timed(lambda: stft(y, n_fft=n_fft, hop_length=hop))
    print(f"n_fft {n_fft:5d} hop {hop:5d}: mel {a:8.3f} ms  {frames / a / 1e3:8.1f} frames/us   "
          f"stft {b:8.3f} ms  {frames / b / 1e3:8.1f} frames/us   (n log n per frame-ns: "
          f"{n_fft * torch.log2(torch.tensor(float(n_fft))).item() * frames / a / 1e6:8.1f})")
