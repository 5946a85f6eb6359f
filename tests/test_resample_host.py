"""Sample-rate conversion without a GPU: the oracle against SciPy (the library the
reference delegates to, resample.py:97,275), the host-side filter design of the product
against SciPy's, and the argument checks of the reference (resample.py:69-81, 245-254)."""

import importlib

import numpy as np
import pytest
import scipy.signal as ss

import oracle.resample as orc
from mlx_audio_amd.primitives import resample, resample_poly

rs_mod = importlib.import_module("mlx_audio_amd.primitives.resample")  # the module, not the function

RATIOS = [(1, 2), (2, 1), (3, 1), (1, 3), (147, 160), (160, 147), (160, 441), (5, 7), (2, 3)]


@pytest.mark.parametrize("up,down", RATIOS)
def test_restated_polyphase_arithmetic_is_scipys(up, down):
    rng = np.random.default_rng(up * 1000 + down)
    for n in (1, 2, 37, 700):
        x = rng.standard_normal(n).astype(np.float32)
        want = ss.resample_poly(x, up, down)
        got = orc.upfirdn_f32(x, up, down)
        assert want.dtype == np.float32
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("up,down", RATIOS + [(1, 6), (441, 160), (320, 441)])
def test_filter_design_matches_scipy_firwin(up, down):
    h, half_len = rs_mod._kaiser_lowpass(up, down)
    m = max(up, down)
    want = ss.firwin(2 * 10 * m + 1, 1.0 / m, window=("kaiser", 5.0)).astype(np.float32)
    want *= up
    assert half_len == 10 * m
    np.testing.assert_array_equal(h, want)
    ho, hl = orc.kaiser_lowpass(up, down)
    np.testing.assert_array_equal(ho, want)
    assert hl == half_len


@pytest.mark.parametrize("up,down", RATIOS)
@pytest.mark.parametrize("n_in", [1, 5, 159, 160, 161, 4410, 100003])
def test_layout_and_phase_table_match_the_oracle(up, down, n_in):
    h, half_len = rs_mod._kaiser_lowpass(up, down)
    got = rs_mod._poly_layout(n_in, up, down, half_len, len(h))
    want = orc.poly_layout(n_in, up, down, half_len, len(h))
    assert got == want
    assert got[3] == len(ss.resample_poly(np.zeros(n_in, np.float32), up, down))
    pre, post = got[0], got[1]
    hp = np.concatenate((np.zeros(pre, np.float32), h, np.zeros(post, np.float32)))
    np.testing.assert_array_equal(rs_mod._phase_table(h, pre, post, up), orc.phase_table(hp, up))


def test_oracle_linear_known_answers():
    # a ramp stays a ramp; the end points are kept (numpy.linspace includes both ends)
    x = np.arange(11, dtype=np.float32)
    y = orc.resample(x, 10, 20, res_type="linear")
    assert y.shape == (22,)
    np.testing.assert_allclose(y, np.linspace(0, 10, 22), rtol=0, atol=1e-6)
    assert y[0] == 0 and y[-1] == 10
    ys = orc.resample(x, 10, 5, res_type="linear", scale=True)
    np.testing.assert_allclose(ys, 0.5 * np.linspace(0, 10, 6)[: len(ys)], atol=1e-6)


def test_oracle_fft_is_scipy_resample():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 441)).astype(np.float32)
    y = orc.resample(x, 44100, 16000)
    assert y.shape == (2, 160) and y.dtype == np.float32
    np.testing.assert_array_equal(y, ss.resample(x, 160, axis=-1).astype(np.float32))


def test_argument_checks_follow_the_reference():
    x = np.zeros(16, np.float32)
    with pytest.raises(ValueError, match="orig_sr must be positive"):
        resample(x, 0, 16000)
    with pytest.raises(ValueError, match="target_sr must be positive"):
        resample(x, 16000, -1)
    with pytest.raises(ValueError, match="up must be positive"):
        resample_poly(x, 0, 1)
    with pytest.raises(ValueError, match="down must be positive"):
        resample_poly(x, 1, 0)
    with pytest.raises(ValueError, match="Unknown res_type: 'sinc'"):
        resample(x, 16000, 8000, res_type="sinc")
    with pytest.raises(ValueError, match="padtype must be one of"):
        resample_poly(x, 1, 2, padtype="nope")
    # the oracle raises the same
    with pytest.raises(ValueError, match="Unknown res_type: 'sinc'"):
        orc.resample(x, 16000, 8000, res_type="sinc")
    with pytest.raises(ValueError, match="up must be positive"):
        orc.resample_poly(x, 0, 1)


def test_identity_cases_return_the_input_object():
    x = np.zeros(16, np.float32)
    assert resample(x, 16000, 16000) is x           # resample.py:73-74
    assert resample_poly(x, 3, 3) is x              # resample.py:249-254
    assert resample_poly(x, 160, 160, padtype="line") is x


def test_no_cpu_fallback_for_any_method():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    for kind in ("fft", "linear"):
        with pytest.raises(RuntimeError, match="needs a ROCm GPU"):
            resample(np.zeros(16, np.float32), 44100, 16000, res_type=kind)
    with pytest.raises(RuntimeError, match="needs a ROCm GPU"):
        resample_poly(np.zeros(16, np.float32), 1, 2)


def test_oracle_reproduces_the_golden_vectors():
    """The committed vectors are SciPy's outputs; the oracle must still produce them here."""
    import os

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "resample_golden.npz"))
    x = gold["x"]
    np.testing.assert_array_equal(orc.resample_poly(x, 160, 441), gold["poly_160_441"])
    np.testing.assert_array_equal(orc.resample_poly(x, 3, 1), gold["poly_3_1"])
    np.testing.assert_array_equal(orc.resample_poly(x, 160, 441, padtype="reflect"),
                                  gold["poly_160_441_reflect"])
    np.testing.assert_array_equal(orc.resample(x, 44100, 16000, res_type="linear"),
                                  gold["linear_44100_16000"])
    np.testing.assert_allclose(orc.resample(x, 44100, 16000), gold["fft_44100_16000"], rtol=0, atol=1e-6)
    for row in range(2):  # and the spelled-out float32 arithmetic gives the same numbers
        np.testing.assert_array_equal(orc.upfirdn_f32(x[row], 160, 441), gold["poly_160_441"][row])
