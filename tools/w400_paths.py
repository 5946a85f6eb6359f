"""Times of the n_fft = 400 / hop 160 front-end on and off its fast kernel (1024 x 30 s clips,
one MI355X): the headline configuration, odd clip lengths, a filterbank other than Whisper's
two, the dB epilogue and the plain mel output.  Prints ms per call and the fraction of the
8 TB/s roofline on the call's algorithmic bytes.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mlx_audio_amd.frontends import compute_log_mel_spectrogram  # noqa: E402
from mlx_audio_amd.primitives import melspectrogram, mfcc  # noqa: E402


def timed(fn, n=30):
    for _ in range(8):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


def report(name, y, fn):
    ms, out = timed(fn)
    outs = out if isinstance(out, (tuple, list)) else [out]
    byts = y.numel() * 4 + sum(o.numel() * 4 for o in outs)
    print(f"{name:78s} {ms:7.3f} ms  {byts / (ms * 1e-3) / 1e9 / 8000 * 100:5.1f} % of 8 TB/s")


B = 1024
_w = torch.randn((B, 480000), device="cuda")
timed(lambda: compute_log_mel_spectrogram(_w, n_mels=128, per_clip=True), 60)  # bring the clock up
del _w
for T in (480000, 480001, 480002, 479999):
    y = torch.randn((B, T), device="cuda")
    for n_mels in (128, 80):
        report(f"whisper log-mel, {n_mels} mels, T = {T} (LDS-DMA kernel)", y,
               lambda: compute_log_mel_spectrogram(y, n_mels=n_mels, per_clip=True))
    del y
y = torch.randn((B, 480000), device="cuda")
kw = dict(sr=16000, n_fft=400, hop_length=160)
report("mel power, Whisper's 128-filter bank, no log (register kernel, table sweep)", y,
       lambda: melspectrogram(y, n_mels=128, fmin=0.0, fmax=8000.0, **kw))
report("mel power, 64 filters 20..7600 Hz (register kernel, table sweep)", y,
       lambda: melspectrogram(y, n_mels=64, fmin=20.0, fmax=7600.0, **kw))
report("mel power, 128 filters, htk scale (register kernel, table sweep)", y,
       lambda: melspectrogram(y, n_mels=128, htk=True, **kw))
report("mfcc 20 of 128 mels at n_fft 400 (register kernel, dB epilogue + DCT)", y,
       lambda: mfcc(y, n_mfcc=20, n_mels=128, per_clip=True, **kw))
