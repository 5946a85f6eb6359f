"""Effective TB/s (bytes in + bytes out over time) of the stand-alone operators around the path, ~1-2 GB each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd.primitives import (amplitude_to_db, db_to_power, db_to_amplitude, magnitude, phase,
                                      zero_crossing_rate, resample, resample_poly, onset_strength_multi)
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def line(name, fn, nbytes):
    try:
        t = timed(fn)
        print(f"{name:42s} {t:8.3f} ms  {nbytes / t / 1e9:6.2f} TB/s")
    except Exception as e:
        print(f"{name:42s} failed: {type(e).__name__}: {e}")
Z = torch.randn((256, 1025, 938), dtype=torch.complex64, device="cuda")
nz = Z.numel()
line("magnitude (complex64 -> f32)", lambda: magnitude(Z), nz * 12)
line("phase", lambda: phase(Z), nz * 12)
Zt = torch.randn((256, 938, 1025), dtype=torch.complex64, device="cuda").transpose(1, 2)
line("magnitude of an stft-layout view", lambda: magnitude(Zt), nz * 12)
line("phase of an stft-layout view", lambda: phase(Zt), nz * 12)
del Z, Zt
S = torch.rand((512, 1025, 938), device="cuda") + 0.01
ns = S.numel()
line("amplitude_to_db", lambda: amplitude_to_db(S), ns * 8)
line("amplitude_to_db top_db=None", lambda: amplitude_to_db(S, top_db=None), ns * 8)
D = amplitude_to_db(S)
line("db_to_power", lambda: db_to_power(D), ns * 8)
line("db_to_amplitude", lambda: db_to_amplitude(D), ns * 8)
del D
M = S[:, :128].contiguous()
line("onset_strength_multi (4 channels)", lambda: onset_strength_multi(S=M, sr=22050, channels=[0, 32, 64, 96, 128]), M.numel() * 4)
del S, M
y = torch.randn((1024, 480000), device="cuda")
ny = y.numel()
line("zero_crossing_rate 2048/512", lambda: zero_crossing_rate(y), ny * 4)
line("zero_crossing_rate 400/160", lambda: zero_crossing_rate(y, frame_length=400, hop_length=160), ny * 4)
line("resample 16k->8k linear", lambda: resample(y, 16000, 8000, method="linear"), ny * 6)
line("resample_poly 2/1", lambda: resample_poly(y[:512], 2, 1), ny // 2 * 12)
line("resample_poly 1/3", lambda: resample_poly(y, 1, 3), ny * 4 * 4 // 3)
line("resample_poly 160/147", lambda: resample_poly(y[:512], 160, 147), ny // 2 * 4 * (1 + 160 / 147))
