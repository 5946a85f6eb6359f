# Times the any-length plans (n_fft without a compiled kernel) and a few Fourier resampling
# length pairs: python tools/time_any_fft.py   (prints ms per call; used to A/B library builds)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import resample, stft


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


y = 0.1 * torch.randn((64, 16000 * 30), device="cuda")
for n_fft, hop in ((1000, 250), (1200, 300), (1536, 384), (6000, 1500), (1001, 250)):
    print(f"stft n_fft {n_fft}: {timed(lambda: stft(y, n_fft=n_fft, hop_length=hop)):.3f} ms")
for rows, sec, a, b in ((64, 30, 44100, 16000), (64, 30, 48000, 16000), (64, 10, 22050, 16000), (1, 30, 44100, 16000),
                        (16, 60, 16000, 44100), (64, 30, 32000, 16000)):
    x = 0.1 * torch.randn((rows, a * sec), device="cuda")
    print(f"resample {rows} x {sec} s {a} -> {b}: {timed(lambda: resample(x, a, b)):.3f} ms")
