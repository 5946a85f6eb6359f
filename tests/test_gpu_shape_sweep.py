"""Seeded random shapes through the kernels whose tiling depends on the shape (the streaming DCT,
the roll-off checkpoint scan, flatness, the zero-crossing span kernel): tiny, ragged and odd
sizes against SciPy / the oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
import scipy.fft  # noqa: E402

import oracle.features as orc  # noqa: E402

pytestmark = pytest.mark.gpu


def test_dct_shapes():
    from mlx_audio_amd.primitives import dct

    rng = np.random.default_rng(0)
    for _ in range(80):
        outer, n_in, inner = int(rng.integers(1, 5)), int(rng.integers(1, 300)), int(rng.integers(1, 400))
        n = int(rng.integers(1, min(n_in, 60) + 1))
        x = rng.standard_normal((outer, n_in, inner)).astype(np.float32)
        got = dct(torch.from_numpy(x).cuda(), n=n, axis=1).cpu().numpy()
        want = scipy.fft.dct(x.astype(np.float64), type=2, norm="ortho", axis=1)[:, :n]
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), (outer, n_in, inner, n)


def test_rolloff_and_flatness_shapes():
    from mlx_audio_amd.primitives import spectral_flatness, spectral_rolloff

    rng = np.random.default_rng(1)
    for it in range(80):
        rows, bins, frames = int(rng.integers(1, 4)), int(rng.integers(1, 700)), int(rng.integers(1, 300))
        S = (rng.standard_normal((rows, bins, frames)) ** 2).astype(np.float32)
        if it % 5 == 0:
            S[:, :, 0] = 0
        freq = np.sort(rng.uniform(0, 8000, bins)).astype(np.float32)
        roll = float(rng.uniform(0, 1))
        got = spectral_rolloff(S=torch.from_numpy(S).cuda(), freq=freq, roll_percent=roll).cpu().numpy()
        want = orc.spectral_rolloff(S=S, freq=freq, roll_percent=roll)
        assert got.shape == want.shape
        assert (got != want).mean() <= 0.05, (rows, bins, frames, roll)   # ties of running sum and threshold
        flat = spectral_flatness(S=torch.from_numpy(S + 1e-3).cuda()).cpu().numpy()
        np.testing.assert_allclose(flat, orc.spectral_flatness(S=S + 1e-3), rtol=2e-4)


def test_zero_crossing_shapes():
    from mlx_audio_amd.primitives import zero_crossing_rate

    rng = np.random.default_rng(2)
    for _ in range(80):
        rows, T = int(rng.integers(1, 4)), int(rng.integers(1, 70000))
        L, hop = int(rng.integers(1, 5000)), int(rng.integers(1, 3000))
        center = bool(rng.integers(0, 2))
        if not center and L > T:
            continue
        y = rng.standard_normal((rows, T)).astype(np.float32)
        y[:, ::11] = 0
        mode = ["edge", "constant"][int(rng.integers(0, 2))]
        got = zero_crossing_rate(torch.from_numpy(y).cuda(), L, hop, center, mode).cpu().numpy()
        want = orc.zero_crossing_rate(y, L, hop, center, mode)
        assert got.shape == want.shape
        np.testing.assert_array_equal(got, want, err_msg=str((rows, T, L, hop, center, mode)))
