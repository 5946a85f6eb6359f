"""MFCC at Whisper's framing (n_fft 400, hop 160, 128 mels -> 20 coefficients; row-major DCT), 1024 x 30 s: ms per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import dct, mfcc
y = torch.randn((1024, 480000), device="cuda")
S = torch.randn((1024, 128, 3001), device="cuda")
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(f"mfcc n_fft 400: {timed(lambda: mfcc(y, sr=16000, n_mfcc=20, n_fft=400, hop_length=160, n_mels=128, fmax=8000.0, per_clip=True)):.3f} ms")
try:
    print(f"dct 128 -> 20 over (1024, 128, 3001): {timed(lambda: dct(S, n=20, axis=1) if True else None):.3f} ms")
except Exception as e:
    print("dct call:", e)
