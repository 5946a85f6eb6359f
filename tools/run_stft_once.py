"""A few stft calls at one framing, for counter passes: python tools/run_stft_once.py N_FFT HOP WIN SR [istft]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import stft, istft
n_fft, hop, win, sr = (int(a) for a in sys.argv[1:5])
y = torch.randn((64, 30 * sr), device="cuda")
C = stft(y, n_fft=n_fft, hop_length=hop, win_length=win)
for _ in range(3):
    if len(sys.argv) > 5:
        istft(C, hop_length=hop, win_length=win, n_fft=n_fft, length=y.shape[1])
    else:
        stft(y, n_fft=n_fft, hop_length=hop, win_length=win)
torch.cuda.synchronize()
