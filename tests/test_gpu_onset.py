"""Onset strength on the GPU against the oracle, through the C ABI.  Tolerance 2e-5 absolute on
envelope values of order 1-10: float32 log1p and a different summation order over the bands;
max and median select values, so they agree to the rounding of log1p alone."""

import numpy as np
import pytest
import torch

import oracle.onset as orc
from mlx_audio_amd import _lib
from mlx_audio_amd.primitives import onset_detect, onset_strength, onset_strength_multi

pytestmark = pytest.mark.gpu

ATOL = 2e-5


def _dev(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


@pytest.mark.parametrize("shape", [(128, 87), (3, 128, 87), (2, 40, 301), (1, 7, 5), (2, 129, 64)])
@pytest.mark.parametrize("aggregate", ["mean", "max", "median"])
def test_flux_from_a_spectrogram(gpu, shape, aggregate):
    rng = np.random.default_rng(sum(shape))
    S = np.abs(rng.standard_normal(shape)).astype(np.float32)
    for lag in (1, 2, 4):
        got = onset_strength(S=_dev(S, gpu), lag=lag, aggregate=aggregate).cpu().numpy()
        want = orc.onset_strength(S=S, lag=lag, aggregate=aggregate)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=ATOL, err_msg=f"lag={lag}")


@pytest.mark.parametrize("kw", [{"max_size": 2}, {"max_size": 3}, {"max_size": 8}, {"detrend": True},
                                {"max_size": 5, "detrend": True, "lag": 3},
                                {"max_size": 400}])
def test_envelope_filters(gpu, kw):
    rng = np.random.default_rng(17)
    S = np.abs(rng.standard_normal((2, 64, 233))).astype(np.float32)
    got = onset_strength(S=_dev(S, gpu), **kw).cpu().numpy()
    want = orc.onset_strength(S=S, **kw)
    np.testing.assert_allclose(got, want, rtol=0, atol=ATOL)


@pytest.mark.parametrize("frames,lag", [(3, 5), (1, 2), (4, 4), (5, 4)])
def test_fewer_frames_than_the_lag(gpu, frames, lag):
    """With no more frames than `lag` both slices of the lagged difference are empty and the
    closing pad leaves `lag` zeros (reference onset.py:131,170-171): the envelope has `lag`
    frames then, whatever the aggregate and the filters; one frame more gives `frames`."""
    rng = np.random.default_rng(frames + 10 * lag)
    S = np.abs(rng.standard_normal((2, 16, frames))).astype(np.float32)
    for aggregate in ("mean", "median", "max"):
        for kw in ({}, {"max_size": 3}, {"detrend": True}):
            want = orc.onset_strength(S=S, lag=lag, aggregate=aggregate, **kw)
            got = onset_strength(S=_dev(S, gpu), lag=lag, aggregate=aggregate, **kw).cpu().numpy()
            assert got.shape == want.shape == (2, max(frames, lag)), (aggregate, kw)
            np.testing.assert_allclose(got, want, rtol=0, atol=ATOL)
    one = onset_strength(S=_dev(S[0], gpu), lag=lag).cpu().numpy()
    assert one.shape == (max(frames, lag),)


@pytest.mark.parametrize("n_fft,hop,n_mels", [(2048, 512, 128), (1024, 256, 64)])
def test_from_a_waveform_and_detection(gpu, n_fft, hop, n_mels):
    sr = 22050
    rng = np.random.default_rng(n_fft)
    y = 0.01 * rng.standard_normal(2 * sr).astype(np.float32)
    clicks = np.arange(4000, 2 * sr - 4000, 7000)
    for c in clicks:
        y[c:c + 400] += np.hanning(400).astype(np.float32) * np.sin(np.arange(400) * 0.9).astype(np.float32)
    kw = dict(n_fft=n_fft, hop_length=hop, n_mels=n_mels)
    got = onset_strength(_dev(y, gpu), sr=sr, **kw).cpu().numpy()
    want = orc.onset_strength(y, sr=sr, **kw)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-3)   # mel values themselves differ ~1e-6 relative
    frames = onset_detect(_dev(y, gpu), sr=sr, hop_length=hop, n_fft=n_fft, n_mels=n_mels).numpy()
    np.testing.assert_array_equal(frames, orc.onset_detect(onset_envelope=got, hop_length=hop))
    assert len(frames) > 0
    # the envelope peaks where the clicks start (within two frames)
    for c in clicks:
        f0 = int(round(c / hop))
        assert got[max(0, f0 - 2):f0 + 3].max() >= 0.5 * got.max()
    yb = np.stack([y, y[::-1].copy()])
    gb = onset_strength(_dev(yb, gpu), sr=sr, **kw).cpu().numpy()
    np.testing.assert_allclose(gb, orc.onset_strength(yb, sr=sr, **kw), rtol=0, atol=2e-3)


def test_multi_channel_keeps_the_reference_behaviour(gpu):
    rng = np.random.default_rng(3)
    y = rng.standard_normal(22050).astype(np.float32)
    multi = onset_strength_multi(_dev(y, gpu), sr=22050)
    assert multi.shape == (3, 44)
    single = onset_strength(_dev(y, gpu), sr=22050)
    # a given spectrogram ignores (fmin, fmax): the three channels are the full-band envelope
    for c in range(3):
        assert torch.equal(multi[c], single)
    assert onset_strength_multi(_dev(np.stack([y, y]), gpu), sr=22050, channels=[(0, 100), (100, 900)]).shape == (2, 2, 44)


def test_c_abi_checks_and_limits(gpu):
    lib = _lib.lib()
    S = torch.ones((4, 6), device=gpu)
    out = torch.zeros(6, device=gpu)
    s = torch.cuda.current_stream(gpu).cuda_stream
    assert lib.mia_onset_flux(S.data_ptr(), 1, 4, 6, 0, 0, out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_onset_flux(S.data_ptr(), 1, 4, 6, 1, 5, out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_onset_flux(S.data_ptr(), 1, 300, 6, 1, 2, out.data_ptr(), s) == _lib.MIA_ERR_UNSUPPORTED
    assert lib.mia_onset_flux(S.data_ptr(), 0, 4, 6, 1, 0, out.data_ptr(), s) == _lib.MIA_OK
    assert lib.mia_envelope_filter(out.data_ptr(), 1, 6, 6, 3, 0, out.data_ptr(), 6, s) == _lib.MIA_ERR_INVALID
    assert lib.mia_envelope_filter(out.data_ptr(), 1, 6, 4, 3, 0, S.data_ptr(), 6, s) == _lib.MIA_ERR_INVALID
    with pytest.raises(ValueError, match="Unknown aggregate"):
        onset_strength(S=S, aggregate="sum")


def test_full_size_batch_properties(gpu):
    """512 mel spectrograms of 30 s (128 bands x 938 frames): spot rows against the oracle;
    the envelope of a spectrogram that never rises is zero; batch invariance bit for bit."""
    g = torch.Generator(device=gpu)
    g.manual_seed(5)
    S = torch.rand((512, 128, 938), generator=g, device=gpu)
    env = onset_strength(S=S)
    rows = [0, 200, 511]
    np.testing.assert_allclose(env[rows].cpu().numpy(), orc.onset_strength(S=S[rows].cpu().numpy()),
                               rtol=0, atol=ATOL)
    assert torch.equal(onset_strength(S=S[200]), env[200])
    falling = torch.cumsum(-torch.rand((4, 128, 938), generator=g, device=gpu), dim=-1) + 1000.0
    assert onset_strength(S=falling).abs().max().item() == 0.0
