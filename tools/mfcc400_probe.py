"""MFCC at Whisper's framing (n_fft 400, hop 160, 128 mels -> 20 coefficients), 1024 x 30 s: for a kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import mfcc
y = torch.randn((1024, 480000), device="cuda")
for _ in range(6):
    out = mfcc(y, sr=16000, n_mfcc=20, n_fft=400, hop_length=160, n_mels=128, fmax=8000.0, per_clip=True)
torch.cuda.synchronize()
print(out.shape)
