"""Whisper log-mel (128 mels, per-clip maximum) over the same 8.5 hours of audio cut into clips of different lengths: ms per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.frontends import compute_log_mel_spectrogram
from mlx_audio_amd.primitives import melspectrogram
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
total = 1024 * 480000
for T in (480000 * 8, 480000, 160000, 48000, 16000, 8000, 4000, 1600):
    B = total // T
    y = torch.randn((B, T), device="cuda")
    a = timed(lambda: compute_log_mel_spectrogram(y, n_mels=128, per_clip=True))
    b = timed(lambda: melspectrogram(y, sr=16000, n_fft=1024, hop_length=256, n_mels=80))
    print(f"{B:7d} clips x {T / 16000:7.2f} s: whisper log-mel {a:7.3f} ms   mel 1024/256/80 {b:7.3f} ms")
    del y
