# istft() on the two physical layouts of a (B, bins, frames) spectrum: a view of (B, frames, bins)
# rows (what stft() returns) and a contiguous (B, bins, frames) array (what a caller who built the
# spectrum elsewhere holds): python tools/time_istft_layouts.py
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import istft, stft


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for n_fft, hop, rows, sec, sr in ((4096, 1024, 32, 60, 44100), (2048, 512, 64, 10, 22050), (1024, 256, 64, 30, 16000), (400, 160, 64, 30, 16000)):
    y = 0.1 * torch.randn((rows, sr * sec), device="cuda")
    S = stft(y, n_fft=n_fft, hop_length=hop)
    Sc = S.contiguous()
    a = timed(lambda: istft(S, hop_length=hop, length=y.shape[1]))
    b = timed(lambda: istft(Sc, hop_length=hop, length=y.shape[1]))
    err = (istft(S, hop_length=hop, length=y.shape[1]) - istft(Sc, hop_length=hop, length=y.shape[1])).abs().max().item()
    print(f"n_fft {n_fft} hop {hop}, {rows} x {sec} s: frames-major view {a:.3f} ms, bins-major contiguous {b:.3f} ms, max diff {err:.1e}")
