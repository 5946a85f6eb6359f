"""Host-side cost of small calls: one 30 s clip (and one 1 s clip) per call, ms per call including the synchronise."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.frontends import clap_log_mel, compute_log_mel_spectrogram
from mlx_audio_amd.primitives import istft, melspectrogram, mfcc, resample_poly, stft

def timed(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        fn(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t1) / n * 1e3

y30 = torch.randn(480000, device="cuda"); y1 = torch.randn(16000, device="cuda")
S = stft(y30, n_fft=1024, hop_length=256)
for name, fn in (("whisper log-mel 1 x 30 s", lambda: compute_log_mel_spectrogram(y30, n_mels=128)),
                 ("whisper log-mel 1 x 1 s", lambda: compute_log_mel_spectrogram(y1, n_mels=80)),
                 ("melspectrogram 2048/512 1 x 30 s", lambda: melspectrogram(y30, sr=16000, n_fft=2048, hop_length=512)),
                 ("mfcc 1 x 30 s", lambda: mfcc(y30, sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128)),
                 ("stft 1024/256 1 x 30 s", lambda: stft(y30, n_fft=1024, hop_length=256)),
                 ("istft 1024/256 1 x 30 s", lambda: istft(S, hop_length=256, length=480000)),
                 ("clap_log_mel 1 x 10 s", lambda: clap_log_mel(y30)),
                 ("resample_poly 48k->16k 1 x 10 s", lambda: resample_poly(y30, 16000, 48000))):
    a, b = timed(fn)
    print(f"{name:36s} back-to-back {a:7.3f} ms/call   with a sync per call {b:7.3f} ms/call")
