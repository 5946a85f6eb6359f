"""Time the one-launch gate + inverse STFT against the three launches it replaces, per window shape."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sg = importlib.import_module("mlx_audio_amd.primitives.spectral_gate")
from mlx_audio_amd.primitives import istft, stft  # noqa: E402


def timed(fn, n=20):
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
    rows, n = 64, 220500
    n_fft = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    hop = n_fft // 4
    x = torch.randn(rows, n, device="cuda") * 0.05
    C = stft(x, n_fft=n_fft, hop_length=hop)
    ops = sg._Ops(C.device)
    sg._FUSED_MAX_FRAMES = sg._FUSED_MAX_FRAMES_4096 = 8   # time every window the kernel takes
    thr = torch.rand(rows, C.shape[1], device="cuda") * 2.0
    print(f"n_fft {n_fft}, {rows} x {n} samples, {C.shape[2]} frames", flush=True)
    print(f"plain istft of the spectrum: {timed(lambda: istft(C, hop_length=hop, length=n)):.3f} ms", flush=True)
    for windows in [(1, 1, 1), (9, 1, 1), (1, 2, 1), (1, 1, 4), (9, 2, 4), (4, 2, 4), (12, 2, 4), (9, 2, 5), (9, 2, 6), (9, 2, 7), (9, 2, 8), (16, 4, 8), (33, 4, 8)]:
        steps = [(a, s) for a, s in zip("fft", windows) if s > 1]
        if ops.gated_istft(C, thr, 0.1, windows, 1.0, n_fft, hop, n_fft, "hann", n) is None:
            print(f"windows {windows}: no one-launch kernel", flush=True)
            continue
        fused = timed(lambda: ops.gated_istft(C, thr, 0.1, windows, 1.0, n_fft, hop, n_fft, "hann", n))
        plain = timed(lambda: istft(ops.gate(None, C, thr, 0.1, steps, False, 1.0), hop_length=hop, length=n))
        print(f"windows {windows}: one launch {fused:.3f} ms, three launches {plain:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
