"""Onset functions without a GPU: closed-form cases for the oracle, the host-side peak picking
of the mirror against the oracle's, and the argument checks (reference onset.py:99-105,
147-148, 250-255, 319-320)."""

import numpy as np
import pytest

import oracle.onset as orc
from mlx_audio_amd.primitives import onset_detect, onset_strength


def test_flux_of_a_step_and_padding():
    S = np.zeros((4, 10), np.float32)
    S[:, 5:] = 1.0                                   # every band switches on at frame 5
    env = orc.onset_strength(S=S)
    assert env.shape == (10,)
    want = np.zeros(10, np.float32)
    want[5] = np.log1p(np.float32(1e4))
    np.testing.assert_allclose(env, want, rtol=1e-6)
    env2 = orc.onset_strength(S=S, lag=2)
    np.testing.assert_allclose(env2[[5, 6]], want[5], rtol=1e-6)
    assert env2[:5].max() == 0 and env2[7:].max() == 0
    # aggregation: only band 0 switches -> mean = 1/4 of it, max = all of it, median = 0
    S1 = np.zeros((4, 10), np.float32)
    S1[0, 5:] = 1.0
    np.testing.assert_allclose(orc.onset_strength(S=S1)[5], want[5] / 4, rtol=1e-6)
    np.testing.assert_allclose(orc.onset_strength(S=S1, aggregate="max")[5], want[5], rtol=1e-6)
    assert orc.onset_strength(S=S1, aggregate="median")[5] == 0.0
    # local max filter widens the peak (window [i-1, i+1] for size 3)
    np.testing.assert_allclose(orc.onset_strength(S=S, max_size=3)[[4, 5, 6]], want[5], rtol=1e-6)
    assert orc.onset_strength(S=S[None]).shape == (1, 10)


def test_peak_picking_mirror_equals_oracle():
    rng = np.random.default_rng(0)
    env = np.abs(rng.standard_normal(300)).astype(np.float32)
    env[::25] += 3.0
    for kw in ({}, {"pre_max": 3, "post_max": 3, "pre_avg": 5, "post_avg": 5, "delta": 0.2, "wait": 4},
               {"backtrack": True}, {"units": "samples"}, {"units": "time"}):
        got = onset_detect(onset_envelope=env, **kw).numpy()
        want = orc.onset_detect(onset_envelope=env, **kw)
        if kw.get("units") == "time":  # float64 on the host, float32 once it is a device array type
            assert got.dtype == np.float32
            want = want.astype(np.float32)
        np.testing.assert_array_equal(got, want)
    assert set(range(0, 300, 25)) <= set(onset_detect(onset_envelope=env).numpy().tolist())
    # batched envelopes: first row only, as in the reference
    np.testing.assert_array_equal(onset_detect(onset_envelope=np.stack([env, env[::-1]])).numpy(),
                                  orc.onset_detect(onset_envelope=env))


def test_argument_checks_follow_the_reference():
    with pytest.raises(ValueError, match="lag must be positive"):
        onset_strength(np.zeros(4096, np.float32), lag=0)
    with pytest.raises(ValueError, match="max_size must be positive"):
        onset_strength(np.zeros(4096, np.float32), max_size=0)
    with pytest.raises(ValueError, match="Either y"):
        onset_strength()
    with pytest.raises(ValueError, match="Either y"):
        onset_detect()
    with pytest.raises(ValueError, match="Unknown units: 'beats'"):
        onset_detect(onset_envelope=np.ones(8, np.float32), units="beats")
    with pytest.raises(ValueError, match="Unknown aggregate: 'sum'"):
        orc.onset_strength(S=np.ones((4, 8), np.float32), aggregate="sum")
