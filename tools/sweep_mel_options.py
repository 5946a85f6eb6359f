"""Mel / stft / istft / mfcc calls with non-default options on 256 x 30 s @ 16 kHz: ms per call, to spot an option that
leaves the fast path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import melspectrogram, stft, istft, mfcc, onset_strength, resample
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def line(name, fn):
    try:
        print(f"{name:64s} {timed(fn):8.3f} ms")
    except Exception as e:
        print(f"{name:64s} failed: {type(e).__name__}: {e}")
y = torch.randn((256, 480000), device="cuda")
for n_fft, hop in ((1024, 256), (2048, 512), (400, 160), (512, 128)):
    base = dict(sr=16000, n_fft=n_fft, hop_length=hop)
    line(f"mel {n_fft}/{hop} defaults", lambda: melspectrogram(y, **base))
    line(f"mel {n_fft}/{hop} win_length = n_fft/2", lambda: melspectrogram(y, win_length=n_fft // 2, **base))
    line(f"mel {n_fft}/{hop} center=False", lambda: melspectrogram(y, center=False, **base))
    line(f"mel {n_fft}/{hop} pad_mode=reflect", lambda: melspectrogram(y, pad_mode="reflect", **base))
    line(f"mel {n_fft}/{hop} pad_mode=edge", lambda: melspectrogram(y, pad_mode="edge", **base))
    line(f"mel {n_fft}/{hop} window=hamming", lambda: melspectrogram(y, window="hamming", **base))
    line(f"mel {n_fft}/{hop} power=1", lambda: melspectrogram(y, power=1.0, **base))
    line(f"mel {n_fft}/{hop} power=1.5", lambda: melspectrogram(y, power=1.5, **base))
    line(f"mel {n_fft}/{hop} n_mels=40 htk", lambda: melspectrogram(y, n_mels=40, htk=True, **base))
    line(f"mel {n_fft}/{hop} n_mels=229", lambda: melspectrogram(y, n_mels=229, **base))
    line(f"mel {n_fft}/{hop} fmin=300 fmax=3400", lambda: melspectrogram(y, fmin=300.0, fmax=3400.0, **base))
    line(f"mel {n_fft}/{hop} norm=None", lambda: melspectrogram(y, norm=None, **base))
    line(f"mfcc {n_fft}/{hop} n_mfcc=13", lambda: mfcc(y, sr=16000, n_fft=n_fft, hop_length=hop, n_mfcc=13))
    line(f"mfcc {n_fft}/{hop} lifter=22", lambda: mfcc(y, sr=16000, n_fft=n_fft, hop_length=hop, lifter=22))
    line(f"onset_strength(y) {n_fft}/{hop}", lambda: onset_strength(y, sr=16000, n_fft=n_fft, hop_length=hop))
    line(f"stft {n_fft}/{hop}", lambda: stft(y, n_fft=n_fft, hop_length=hop))
    line(f"stft {n_fft}/{hop} win_length n_fft/2 center=False", lambda: stft(y, n_fft=n_fft, hop_length=hop, win_length=n_fft // 2, center=False))
    line(f"stft {n_fft}/{hop // 2} (hop n_fft/8)", lambda: stft(y[:128], n_fft=n_fft, hop_length=hop // 2))
    C = stft(y, n_fft=n_fft, hop_length=hop)
    line(f"istft {n_fft}/{hop}", lambda: istft(C, hop_length=hop))
    line(f"istft {n_fft}/{hop} length=470000", lambda: istft(C, hop_length=hop, length=470000))
    line(f"istft {n_fft}/{hop} center=False", lambda: istft(C, hop_length=hop, center=False))
    line(f"istft {n_fft}/{hop} window=hamming", lambda: istft(C, hop_length=hop, window="hamming"))
    del C
    C = stft(y, n_fft=n_fft, hop_length=hop // 2 * 3)
    line(f"istft {n_fft}/{hop // 2 * 3} (hop 3 n_fft/8)", lambda: istft(C, hop_length=hop // 2 * 3))
    del C
