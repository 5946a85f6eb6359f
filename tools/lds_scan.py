"""One call of every transform kernel family on a mid-sized batch (for a counter pass over all of them:
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -- python3 tools/lds_scan.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_audio_amd.primitives import istft, melspectrogram, mfcc, stft  # noqa: E402

y = 0.1 * torch.randn((64, 16000 * 30), device="cuda")
for n_fft in (256, 400, 512, 800, 1024, 2048, 4096, 320, 480, 640, 960, 1920, 441, 882, 1000):
    hop = n_fft // 4
    for _ in range(2):
        S = stft(y, n_fft=n_fft, hop_length=hop)
        melspectrogram(y, sr=16000, n_fft=n_fft, hop_length=hop, n_mels=64)
        istft(S, hop_length=hop, length=y.shape[1])
    if n_fft in (400, 1024, 2048):
        mfcc(y, sr=16000, n_fft=n_fft, hop_length=hop, n_mels=64, n_mfcc=20)
torch.cuda.synchronize()
print("done")
