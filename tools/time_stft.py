# Times stft() alone on the HTDemucs-size batch (128 rows x 180 s @44.1 kHz, n_fft 4096 hop 1024).
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import stft
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
y = 0.1 * torch.randn((rows, 44100 * 180), device="cuda")
for _ in range(3):
    S = stft(y, n_fft=4096, hop_length=1024, pad_mode="reflect")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    S = stft(y, n_fft=4096, hop_length=1024, pad_mode="reflect")
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
gb = (y.numel() * 4 + S.numel() * 8) / 1e9
print(f"stft {rows} rows: {ms:.3f} ms, {gb / ms:.2f} TB/s on {gb:.1f} GB")
