"""The oracle of the Whisper encoder stem against torch's conv1d / gelu on the CPU (float64):
an independent implementation of the same definitions (the reference's own arithmetic, MLX,
is absent; it stores no numbers for this stage)."""
import numpy as np
import pytest
import torch

from oracle import whisper_stem as ws


@pytest.mark.parametrize("n_mels,n_state,T", [(80, 128, 50), (128, 256, 37), (16, 128, 1), (8, 128, 2)])
def test_oracle_stem_matches_torch(n_mels, n_state, T):
    rng = np.random.default_rng(n_mels + T)
    mel = rng.standard_normal((2, n_mels, T))
    w1 = rng.standard_normal((n_state, 3, n_mels)) * 0.1
    b1 = rng.standard_normal(n_state) * 0.1
    w2 = rng.standard_normal((n_state, 3, n_state)) * 0.05
    b2 = rng.standard_normal(n_state) * 0.1
    pos = ws.sinusoids((T - 1) // 2 + 1 + 3, n_state).astype(np.float64)
    got = ws.encoder_stem(mel, w1, b1, w2, b2, pos, dtype=np.float64)
    x = torch.from_numpy(mel)
    # torch holds conv weights as (out, in, kernel) and works channels-first
    h = torch.nn.functional.gelu(torch.nn.functional.conv1d(
        x, torch.from_numpy(w1).permute(0, 2, 1).contiguous(), torch.from_numpy(b1), stride=1, padding=1))
    h = torch.nn.functional.gelu(torch.nn.functional.conv1d(
        h, torch.from_numpy(w2).permute(0, 2, 1).contiguous(), torch.from_numpy(b2), stride=2, padding=1))
    want = h.permute(0, 2, 1).numpy() + pos[None, :h.shape[2], :]
    assert got.shape == want.shape == (2, (T - 1) // 2 + 1, n_state)
    assert np.abs(got - want).max() < 1e-11


def test_sinusoids_known_values():
    p = ws.sinusoids(1500, 1280)
    assert p.shape == (1500, 1280) and p.dtype == np.float32
    assert np.all(p[0, :640] == 0.0) and np.all(p[0, 640:] == 1.0)
    assert abs(p[1, 0] - np.sin(1.0)) < 1e-7 and abs(p[1, 640] - np.cos(1.0)) < 1e-7
    # the slowest channel turns at 1 / max_timescale
    assert abs(p[1, 639] - np.sin(1e-4)) < 1e-9


def test_gelu_known_values():
    x = np.array([0.0, 1.0, -1.0, 3.0], dtype=np.float64)
    assert np.allclose(ws.gelu(x), [0.0, 0.8413447460685429, -0.15865525393145707, 2.9959503019541966],
                       atol=1e-15)
