"""Spectral features without a GPU: closed-form cases for the oracle (the reference's own
tests of this module compare against librosa at run time and hold no numbers) and the
argument checks of the mirror (reference features.py:349, 715-716, 732-735, 43-44)."""

import numpy as np
import pytest

import oracle.features as orc
from mlx_audio_amd.primitives import (spectral_centroid, spectral_contrast, spectral_rolloff,
                                      zero_crossing_rate)


def test_centroid_and_bandwidth_of_a_single_line():
    freq = np.linspace(0, 8000, 9).astype(np.float32)
    S = np.zeros((9, 4), np.float32)
    S[3] = 2.0                        # all energy at 3 kHz
    np.testing.assert_allclose(orc.spectral_centroid(S=S, freq=freq), 3000.0, rtol=1e-6)
    np.testing.assert_allclose(orc.spectral_bandwidth(S=S, freq=freq), 0.0, atol=1e-3)
    S[5] = 2.0                        # equal lines at 3 and 5 kHz: centroid 4 kHz, spread 1 kHz
    np.testing.assert_allclose(orc.spectral_centroid(S=S, freq=freq), 4000.0, rtol=1e-6)
    np.testing.assert_allclose(orc.spectral_bandwidth(S=S, freq=freq), 1000.0, rtol=1e-5)
    np.testing.assert_allclose(orc.spectral_bandwidth(S=S, freq=freq, p=1.0), 1000.0, rtol=1e-5)
    np.testing.assert_allclose(orc.spectral_bandwidth(S=S, freq=freq, norm=False),
                               np.sqrt(4.0) * 1000.0, rtol=1e-5)


def test_rolloff_and_flatness_closed_forms():
    freq = np.arange(10, dtype=np.float32) * 100
    S = np.ones((10, 3), np.float32)
    # the running sum reaches 85 % of 10 at the 9th bin (index 8)
    np.testing.assert_array_equal(orc.spectral_rolloff(S=S, freq=freq, roll_percent=0.85), 800.0)
    np.testing.assert_array_equal(orc.spectral_rolloff(S=S, freq=freq, roll_percent=0.0), 0.0)
    np.testing.assert_array_equal(orc.spectral_rolloff(S=S, freq=freq, roll_percent=1.0), 900.0)
    # a flat spectrum has flatness 1, a single line among zeros essentially 0
    np.testing.assert_allclose(orc.spectral_flatness(S=S), 1.0, rtol=1e-6)
    line = np.zeros((10, 3), np.float32)
    line[2] = 1.0
    assert orc.spectral_flatness(S=line).max() < 1e-7
    assert orc.spectral_centroid(S=S, freq=freq).shape == (1, 3)
    assert orc.spectral_centroid(S=S[None], freq=freq).shape == (1, 1, 3)


def test_zero_crossing_rate_closed_forms():
    y = np.tile(np.array([1.0, -1.0], np.float32), 2048)         # alternates every sample
    z = orc.zero_crossing_rate(y, frame_length=256, hop_length=128, center=False)
    np.testing.assert_allclose(z, 255.0 / 256.0)
    assert z.shape == (1, 1 + (4096 - 256) // 128)
    assert orc.zero_crossing_rate(np.ones(4096, np.float32)).max() == 0.0
    # -0.0 counts as non-negative for (x >= 0) and as negative for the sign bit
    y = np.array([1.0, -0.0, 1.0, -0.0] * 64, np.float32)
    assert orc.zero_crossing_rate(y, 256, 256, center=False, use_mlx=True).max() == 0.0
    np.testing.assert_allclose(orc.zero_crossing_rate(y, 256, 256, center=False, use_mlx=False),
                               255.0 / 256.0)


def test_features_of_a_sine_follow_its_frequency():
    sr, f0 = 22050, 2000.0
    y = np.sin(2 * np.pi * f0 * np.arange(sr) / sr).astype(np.float32)
    c = orc.spectral_centroid(y, sr=sr)[0, 2:-2]
    assert np.abs(c - f0).max() < 30.0
    assert np.abs(orc.spectral_rolloff(y, sr=sr)[0, 2:-2] - f0).max() < 30.0
    assert orc.spectral_flatness(y)[0, 2:-2].max() < 1e-4
    np.testing.assert_allclose(orc.zero_crossing_rate(y)[0, 2:-2], 2 * f0 / sr, atol=2e-3)


def test_argument_checks_follow_the_reference():
    y = np.zeros(4096, np.float32)
    with pytest.raises(ValueError, match="roll_percent must be <= 1.0"):
        spectral_rolloff(y, roll_percent=1.5)
    with pytest.raises(ValueError, match="roll_percent must be >= 0.0"):
        spectral_rolloff(y, roll_percent=-0.1)
    with pytest.raises(ValueError, match="frame_length must be positive"):
        zero_crossing_rate(y, frame_length=0)
    with pytest.raises(ValueError, match="hop_length must be positive"):
        zero_crossing_rate(y, hop_length=0)
    with pytest.raises(ValueError, match="Unknown pad_mode: 'reflect'"):
        zero_crossing_rate(y, pad_mode="reflect")
    with pytest.raises(ValueError, match="n_bands must be positive"):
        spectral_contrast(y, n_bands=0)
    with pytest.raises(ValueError, match="quantile must be <= 1.0"):
        spectral_contrast(y, quantile=1.5)
    with pytest.raises(ValueError, match="roll_percent must be <= 1.0"):
        orc.spectral_rolloff(y, roll_percent=1.5)
    with pytest.raises(ValueError, match="Either y"):
        orc.spectral_centroid()


def test_either_y_or_S_is_required():
    import torch

    if torch.cuda.is_available():
        with pytest.raises(ValueError, match="Either y"):
            spectral_centroid()
    else:
        with pytest.raises((ValueError, RuntimeError)):
            spectral_centroid()


def test_contrast_band_table_and_closed_form():
    import importlib

    mod = importlib.import_module("mlx_audio_amd.primitives.features")
    for sr, n_fft, fmin, n_bands, q in [(22050, 2048, 200.0, 6, 0.02), (16000, 512, 100.0, 5, 0.1),
                                        (44100, 4096, 200.0, 6, 0.02), (8000, 256, 200.0, 6, 0.5)]:
        freq = orc.get_frequencies(sr, n_fft)
        table = mod._contrast_bands(freq, fmin, n_bands, q)
        want = orc.contrast_bands(freq, fmin, n_bands, q)
        assert table.shape == (n_bands + 1, 3)
        for row, band in zip(table, want):
            if band is None:
                assert row[1] <= row[0]
            else:
                sel, nq = band
                np.testing.assert_array_equal(np.arange(row[0], row[1]), sel)
                assert row[2] == nq
    # two levels per band: peak = the high level, valley = the low one
    freq = orc.get_frequencies(22050, 2048)
    S = np.full((1025, 3), 0.01, np.float32)
    S[::2] = 1.0
    c = orc.spectral_contrast(S=S, sr=22050, quantile=0.1)
    assert c.shape == (7, 3)
    np.testing.assert_allclose(c, 20.0, atol=1e-4)          # 10 log10(1 / 0.01)
    np.testing.assert_allclose(orc.spectral_contrast(S=S, sr=22050, quantile=0.1, linear=True), 0.99,
                               atol=1e-6)
