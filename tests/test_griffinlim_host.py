"""Griffin-Lim without a GPU: the oracle converges, keeps shapes, and both sides check their
arguments as the reference does (griffinlim.py:91-92, 118-119)."""

import numpy as np
import pytest

import oracle.griffinlim as orc
import oracle.spectral as sp
from mlx_audio_amd.primitives import griffinlim


def _consistency(y, S, n_fft, hop):
    M = np.abs(sp.stft(y, n_fft=n_fft, hop_length=hop))
    return np.linalg.norm(M - S) / np.linalg.norm(S)


def test_oracle_converges_and_is_seeded():
    rng = np.random.default_rng(0)
    y = rng.standard_normal(4096).astype(np.float32)
    S = np.abs(sp.stft(y, n_fft=512, hop_length=128))
    errs = [_consistency(orc.griffinlim(S, n_iter=n, random_state=0), S, 512, 128) for n in (1, 8, 32)]
    assert errs[0] > errs[1] > errs[2] and errs[2] < 0.15
    a = orc.griffinlim(S, n_iter=2, random_state=3)
    np.testing.assert_array_equal(a, orc.griffinlim(S, n_iter=2, random_state=3))
    assert orc.griffinlim(S[None], n_iter=1, random_state=0).shape == (1, 4096)
    assert orc.griffinlim(S, n_iter=1, length=4000, init="zeros").shape == (4000,)
    ang, out, err = orc.griffinlim_iter(S, np.zeros_like(S), 128, 512, 512)
    assert ang.shape == S.shape and out.shape == S.shape and err > 0


def test_argument_checks_follow_the_reference():
    S = np.ones((9, 4), np.float32)
    for fn in (griffinlim, orc.griffinlim):
        with pytest.raises(ValueError, match="n_iter must be positive"):
            fn(S, n_iter=0)
        with pytest.raises(ValueError, match="momentum must be < 1.0"):
            fn(S, momentum=1.0)
        with pytest.raises(ValueError, match="momentum must be >= 0.0"):
            fn(S, momentum=-0.1)
    with pytest.raises(ValueError, match="Unknown init: 'ones'"):
        orc.griffinlim(S, init="ones")
