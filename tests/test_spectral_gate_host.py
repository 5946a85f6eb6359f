"""Spectral gating without a GPU: the oracle attenuates noise and keeps the tone, and the
argument checks of both sides follow the reference (spectral_gate.py:116-118, 295-297)."""

import numpy as np
import pytest

import oracle.spectral_gate as orc
from mlx_audio_amd.primitives import spectral_gate, spectral_gate_adaptive


def _noisy_tone(n=22050, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 22050.0
    tone = np.sin(2 * np.pi * 440 * t) * (t > 0.3)
    return (tone + 0.05 * rng.standard_normal(n)).astype(np.float32), tone.astype(np.float32)


def test_oracle_reduces_noise_and_keeps_the_tone():
    y, tone = _noisy_tone()
    seg = slice(9000, 20000)
    for fn in (lambda: orc.spectral_gate(y), lambda: orc.spectral_gate(y, noise_profile=y[:5000]),
               lambda: orc.spectral_gate(y, stationary=False)):
        out = fn()
        assert out.shape == y.shape and out.dtype == np.float32
        assert np.abs(out[:5000]).mean() < 0.9 * np.abs(y[:5000]).mean()       # the noise-only part
        assert np.corrcoef(out[seg], tone[seg])[0, 1] > 0.99
    # the adaptive variant takes the running minimum over +-100 ms as the noise floor, so a
    # steady tone counts as floor and is gated too: that is the reference's algorithm
    out = orc.spectral_gate_adaptive(y)
    assert out.shape == y.shape and np.abs(out[:5000]).mean() < 0.9 * np.abs(y[:5000]).mean()
    assert np.abs(out[seg]).mean() < 0.1 * np.abs(y[seg]).mean()
    assert orc.spectral_gate(np.stack([y, y])).shape == (2, len(y))
    # prop_decrease = 0 leaves the signal alone (up to the STFT round trip)
    np.testing.assert_allclose(orc.spectral_gate(y, prop_decrease=0.0), y, atol=1e-5)


def test_argument_checks_follow_the_reference():
    y = np.zeros(4096, np.float32)
    for fn in (spectral_gate, spectral_gate_adaptive, orc.spectral_gate, orc.spectral_gate_adaptive):
        with pytest.raises(ValueError, match="n_fft must be positive"):
            fn(y, n_fft=0, hop_length=1)
        with pytest.raises(ValueError, match="hop_length must be positive"):
            fn(y, hop_length=0)
        with pytest.raises(ValueError, match="prop_decrease must be <= 1.0"):
            fn(y, prop_decrease=1.5)
        with pytest.raises(ValueError, match="prop_decrease must be >= 0.0"):
            fn(y, prop_decrease=-0.5)
