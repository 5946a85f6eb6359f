"""Random framings, window shapes and clip shapes through the one-launch gate + inverse STFT against the separate launches."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sg = importlib.import_module("mlx_audio_amd.primitives.spectral_gate")


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    worst, fused_calls = 0.0, 0
    sg._FUSED_MAX_FRAMES = sg._FUSED_MAX_FRAMES_4096 = 8
    for it in range(120):
        n_fft = int(rng.choice([1024, 2048, 4096]))
        rows = int(rng.integers(1, 10))
        n = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 150000)]))
        sr = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
        kw = dict(n_fft=n_fft, sr=sr, n_grad_freq=int(rng.integers(1, 5)), n_grad_time=int(rng.integers(1, 9)),
                  prop_decrease=float(rng.uniform(0, 1)), threshold_db=float(rng.uniform(-40, 0)),
                  freq_smooth_hz=float(rng.choice([0.0, 50.0, 100.0, 150.0])), time_smooth_ms=0.0)
        if rng.random() < 0.3:
            kw["win_length"] = int(rng.integers(n_fft // 2, n_fft + 1))
            kw["window"] = str(rng.choice(["hann", "hamming", "blackman"]))
        x = 0.1 * torch.randn((rows, n), generator=g, device="cuda")
        x[:, n // 3:] *= 5.0
        if rng.random() < 0.3:
            kw["noise_profile"] = 0.1 * torch.randn((int(rng.integers(n_fft, 20000)),), generator=g, device="cuda")
        steps = []
        fb = max(1, int(kw["freq_smooth_hz"] * n_fft / sr))
        steps += [("f", fb)] if fb > 1 else []
        windows = (fb, kw["n_grad_freq"], kw["n_grad_time"])
        sg._FUSE_INVERSE = True
        a = sg.spectral_gate(x, **kw)
        sg._FUSE_INVERSE = False
        b = sg.spectral_gate(x, **kw)
        sg._FUSE_INVERSE = True
        d = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-3)
        fused_calls += int(fb <= 16)
        worst = max(worst, d)
        if d > 1e-6 or not torch.isfinite(a).all():
            print("MISMATCH", it, kw, rows, n, windows, d, flush=True)
    print(f"120 cases ({fused_calls} with a first window the kernel takes), worst relative difference {worst:.3g}")


if __name__ == "__main__":
    main()
