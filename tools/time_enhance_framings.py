"""stft / istft at the framings of the reference's speech-enhancement model (models/enhance/config.py:28-31, 63-68:
n_fft 1920, hop 480, window 960 at 48 kHz; n_fft 640, hop 160, window 320 at 16 kHz), 64 x 30 s: ms per call and TB/s
of signal + spectrum bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import stft, istft
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for sr, n_fft, hop, win in ((48000, 1920, 480, 960), (16000, 640, 160, 320), (48000, 2048, 512, 2048), (16000, 512, 128, 512)):
    y = torch.randn((64, 30 * sr), device="cuda")
    kw = dict(n_fft=n_fft, hop_length=hop, win_length=win)
    C = stft(y, **kw)
    nbytes = y.numel() * 4 + C.numel() * 8
    t1 = timed(lambda: stft(y, **kw))
    t2 = timed(lambda: istft(C, hop_length=hop, win_length=win, n_fft=n_fft, length=y.shape[1]))
    back = istft(C, hop_length=hop, win_length=win, n_fft=n_fft, length=y.shape[1])
    err = (back - y).abs().max().item()
    print(f"n_fft {n_fft:5d} hop {hop:4d} win {win:5d}: stft {t1:.3f} ms {nbytes / t1 / 1e9:.2f} TB/s   istft {t2:.3f} ms {nbytes / t2 / 1e9:.2f} TB/s   round trip max err {err:.2e}")
    del y, C, back
