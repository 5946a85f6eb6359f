# Times the Whisper log-mel step for clip lengths given on the command line
# (default 480000): usage quick_time.py [n_mels] T [T ...]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.frontends import compute_log_mel_spectrogram
args = [int(a) for a in sys.argv[1:]] or [128, 480000]
n_mels, Ts = args[0], args[1:]
for T in Ts:
    y = torch.randn((1024, T), device="cuda")
    for _ in range(20):
        out = compute_log_mel_spectrogram(y, n_mels=n_mels, per_clip=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        out = compute_log_mel_spectrogram(y, n_mels=n_mels, per_clip=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    byts = y.numel() * 4 + out.numel() * 4
    print(f"n_mels={n_mels} T={T} F={out.shape[-1]} ms={ms:.4f} algorithmic TB/s={byts / ms / 1e9:.3f}")
    del y, out
