# Time of the generic forward kernel with and without the mel stage on the MFCC / CLAP shapes.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import stft, melspectrogram

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for name, B, T, sr, n_fft, hop, n_mels in (("mfcc", 256, 480000, 16000, 2048, 512, 128), ("clap", 256, 480000, 48000, 1024, 480, 64)):
    y = torch.randn((B, T), device="cuda")
    a = timeit(lambda: stft(y, n_fft=n_fft, hop_length=hop))
    b = timeit(lambda: melspectrogram(y, sr=sr, n_fft=n_fft, hop_length=hop, n_mels=n_mels))
    print(f"{name}: B={B} stft {a:.3f} ms (incl. output alloc)  melspectrogram {b:.3f} ms")
