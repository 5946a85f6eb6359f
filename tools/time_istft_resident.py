"""Fused inverse STFT reading a spectrum from HBM against the same launch reading one row again and again (L2-resident source)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_audio_amd.primitives.stft import _istft_raw  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    for n_fft, rows, seconds, sr in [(4096, 128, 180, 44100), (2048, 256, 30, 22050), (1024, 256, 30, 16000)]:
        hop = n_fft // 4
        n = seconds * sr
        frames = 1 + n // hop
        bins = n_fft // 2 + 1
        full = torch.randn((rows, frames, bins), dtype=torch.complex64, device="cuda")
        one = full[:1].expand(rows, frames, bins)
        padded = n + n_fft
        t_full = timed(lambda: _istft_raw(full.transpose(1, 2), n_fft, hop, n_fft, "hann", padded))
        t_one = timed(lambda: _istft_raw(one.transpose(1, 2), n_fft, hop, n_fft, "hann", padded))
        gb = (full.numel() * 8 + rows * padded * 4) / 1e9
        print(f"n_fft {n_fft}: {rows} rows x {frames} frames ({gb:.2f} GB): from HBM {t_full:.3f} ms ({gb / t_full:.2f} TB/s), "
              f"one row re-read {t_one:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
