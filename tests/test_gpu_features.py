"""Spectral features on the GPU against the oracle (float32 NumPy restatement of the
reference), through the C ABI.  Tolerances: float32 sums over up to 2049 bins in a different
order (1e-5 relative); the roll-off frequency may land one bin away when the running sum and
the threshold agree to the last bits (the reference itself changes there with the order of
its cumulative sum)."""

import numpy as np
import pytest
import torch

import oracle.features as orc
from mlx_audio_amd import _lib
from mlx_audio_amd.primitives import (fused_statistics, magnitude, stft, spectral_bandwidth, spectral_centroid, spectral_contrast,
                                      spectral_flatness, spectral_rolloff, zero_crossing_rate)

pytestmark = pytest.mark.gpu

RTOL = 2e-5


def _dev(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def _spec(rng, shape):
    return (rng.standard_normal(shape).astype(np.float32) ** 2 + 0.01).astype(np.float32)


@pytest.mark.parametrize("shape", [(1025, 44), (3, 1025, 44), (2, 513, 301), (5, 9, 1), (1, 2049, 130)])
def test_statistics_from_a_spectrogram(gpu, shape):
    rng = np.random.default_rng(sum(shape))
    S = _spec(rng, shape)
    bins = shape[-2]
    sr, n_fft = 22050, 2 * (bins - 1)
    Sd = _dev(S, gpu)
    got = spectral_centroid(S=Sd, sr=sr, n_fft=n_fft).cpu().numpy()
    want = orc.spectral_centroid(S=S, sr=sr, n_fft=n_fft)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=RTOL)
    for p, norm in ((2.0, True), (1.0, True), (3.0, True), (2.0, False)):
        got = spectral_bandwidth(S=Sd, sr=sr, n_fft=n_fft, p=p, norm=norm).cpu().numpy()
        want = orc.spectral_bandwidth(S=S, sr=sr, n_fft=n_fft, p=p, norm=norm)
        np.testing.assert_allclose(got, want, rtol=5e-5, err_msg=f"p={p} norm={norm}")
    got = spectral_flatness(S=Sd).cpu().numpy()
    np.testing.assert_allclose(got, orc.spectral_flatness(S=S), rtol=5e-5)
    freq = orc.get_frequencies(sr, n_fft)
    step = freq[1] - freq[0] if bins > 1 else 0.0
    for roll in (0.85, 0.5, 0.0, 1.0):
        got = spectral_rolloff(S=Sd, sr=sr, n_fft=n_fft, roll_percent=roll).cpu().numpy()
        want = orc.spectral_rolloff(S=S, sr=sr, n_fft=n_fft, roll_percent=roll)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= step * 1.001, roll
        assert (got != want).mean() <= 0.02, roll


def test_custom_frequencies_and_given_centroid(gpu):
    rng = np.random.default_rng(4)
    S = _spec(rng, (2, 64, 20))
    freq = np.sort(rng.uniform(0, 8000, 64)).astype(np.float32)
    c = orc.spectral_centroid(S=S, freq=freq)
    np.testing.assert_allclose(spectral_centroid(S=_dev(S, gpu), freq=freq).cpu().numpy(), c, rtol=RTOL)
    got = spectral_bandwidth(S=_dev(S, gpu), freq=_dev(freq, gpu), centroid=_dev(c, gpu)).cpu().numpy()
    np.testing.assert_allclose(got, orc.spectral_bandwidth(S=S, freq=freq, centroid=c), rtol=5e-5)
    with pytest.raises(ValueError, match="freq has 10 entries"):
        spectral_centroid(S=_dev(S, gpu), freq=freq[:10])


@pytest.mark.parametrize("n_fft,hop", [(2048, 512), (1024, 256), (400, 160)])
def test_statistics_from_a_waveform(gpu, n_fft, hop):
    rng = np.random.default_rng(n_fft)
    t = np.arange(22050) / 22050.0
    y = (np.sin(2 * np.pi * 1500 * t) + 0.1 * rng.standard_normal(22050)).astype(np.float32)
    yb = np.stack([y, rng.standard_normal(22050).astype(np.float32)])
    kw = dict(n_fft=n_fft, hop_length=hop)
    for fn, ofn, tol in ((spectral_centroid, orc.spectral_centroid, 1e-4),
                         (spectral_bandwidth, orc.spectral_bandwidth, 2e-4)):
        got = fn(_dev(yb, gpu), sr=22050, **kw).cpu().numpy()
        want = ofn(yb, sr=22050, **kw)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=tol)
        np.testing.assert_allclose(fn(_dev(y, gpu), sr=22050, **kw).cpu().numpy(),
                                   ofn(y, sr=22050, **kw), rtol=tol)
    got = spectral_flatness(_dev(yb, gpu), **kw).cpu().numpy()
    np.testing.assert_allclose(got, orc.spectral_flatness(yb, **kw), rtol=1e-3, atol=1e-9)
    got = spectral_flatness(_dev(yb, gpu), power=1.0, **kw).cpu().numpy()
    np.testing.assert_allclose(got, orc.spectral_flatness(yb, power=1.0, **kw), rtol=1e-3)
    got = spectral_rolloff(_dev(yb, gpu), sr=22050, **kw).cpu().numpy()
    want = orc.spectral_rolloff(yb, sr=22050, **kw)
    assert np.abs(got - want).max() <= 22050 / n_fft * 1.001
    assert (got != want).mean() <= 0.02


@pytest.mark.parametrize("frame_length,hop", [(2048, 512), (256, 128), (100, 37), (1, 1), (64, 200), (9000, 33),
                                              (300000, 1000), (65, 64), (4096, 4096)])
def test_zero_crossing_rate_is_exact(gpu, frame_length, hop):
    rng = np.random.default_rng(frame_length)
    y = rng.standard_normal((3, 10000)).astype(np.float32)
    y[0, 100:200] = 0.0
    y[1, ::7] = -0.0
    for center in (True, False):
        if not center and frame_length > y.shape[1]:
            continue
        for pad_mode in ("edge", "constant"):
            for use_mlx in (True, False):
                want = orc.zero_crossing_rate(y, frame_length, hop, center, pad_mode, use_mlx)
                got = zero_crossing_rate(_dev(y, gpu), frame_length, hop, center, pad_mode, use_mlx)
                assert got.shape == want.shape
                np.testing.assert_array_equal(got.cpu().numpy(), want)
    np.testing.assert_array_equal(zero_crossing_rate(_dev(y[2], gpu), frame_length, hop).cpu().numpy(),
                                  orc.zero_crossing_rate(y[2], frame_length, hop))


@pytest.mark.parametrize("frame_length,hop", [(2048, 512), (400, 160), (1024, 1000), (50, 7)])
def test_zero_crossing_rate_over_long_rows(gpu, frame_length, hop):
    """Rows long enough that several workgroups share one (their spans overlap by frame_length -
    hop samples) and a span's groups of 64 samples do not line up with the row."""
    rng = np.random.default_rng(hop)
    y = rng.standard_normal((2, 150001)).astype(np.float32)
    y[0, 70000:71000] = 0.0
    y[1, 5::3] *= -0.0
    for center, pad_mode in ((True, "edge"), (True, "constant"), (False, "edge")):
        want = orc.zero_crossing_rate(y, frame_length, hop, center, pad_mode)
        got = zero_crossing_rate(_dev(y, gpu), frame_length, hop, center, pad_mode)
        np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("sr,n_fft,kw", [
    (22050, 2048, {}), (22050, 2048, {"linear": True}), (16000, 512, {"fmin": 100.0, "n_bands": 5, "quantile": 0.1}),
    (44100, 4096, {}), (8000, 256, {"quantile": 0.5}), (22050, 1024, {"quantile": 1.0}),
    (22050, 2048, {"quantile": 0.0}), (22050, 2048, {"quantile": 0.12}), (48000, 1024, {"fmin": 4000.0, "n_bands": 4}),
])
def test_spectral_contrast(gpu, sr, n_fft, kw):
    """Order statistics: the same values are selected, only the order of the few additions in the
    means differs (1e-6 relative on the linear values; 2e-5 dB)."""
    rng = np.random.default_rng(n_fft + sr)
    S = _spec(rng, (2, n_fft // 2 + 1, 37))
    got = spectral_contrast(S=_dev(S, gpu), sr=sr, n_fft=n_fft, **kw).cpu().numpy()
    want = orc.spectral_contrast(S=S, sr=sr, n_fft=n_fft, **kw)
    assert got.shape == want.shape == (2, kw.get("n_bands", 6) + 1, 37)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=3e-5)
    one = spectral_contrast(S=_dev(S[0], gpu), sr=sr, n_fft=n_fft, **kw).cpu().numpy()
    np.testing.assert_allclose(one, want[0], rtol=2e-6, atol=3e-5)


def test_spectral_contrast_from_a_waveform_and_limits(gpu):
    rng = np.random.default_rng(5)
    y = rng.standard_normal((2, 22050)).astype(np.float32)
    got = spectral_contrast(_dev(y, gpu), sr=22050).cpu().numpy()
    want = orc.spectral_contrast(y, sr=22050)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-3)
    S = _spec(rng, (1, 4097, 5))   # 0.2 of the 2800-bin top band is far more than 64 values:
    got = spectral_contrast(S=_dev(S, gpu), sr=44100, n_fft=8192, quantile=0.2).cpu().numpy()
    want = orc.spectral_contrast(S=S, sr=44100, n_fft=8192, quantile=0.2)   # ... bands by sorting
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=3e-5)
    # the C entry point itself still refuses what its register arrays cannot hold
    bands = torch.tensor([[0, 4097, 100]], dtype=torch.int32, device=gpu)
    out = torch.empty((1, 1, 5), device=gpu)
    rc = _lib.lib().mia_spectral_contrast(_dev(S, gpu).data_ptr(), 1, 4097, 5, bands.data_ptr(), 1, 100, 0,
                                          out.data_ptr(), None)
    assert rc == _lib.MIA_ERR_UNSUPPORTED


@pytest.mark.parametrize("n_fft,hop", [(2048, 512), (1024, 256), (400, 160), (512, 128), (4096, 1024), (256, 64), (800, 200)])
def test_fused_statistics_equal_the_spectrogram_path(gpu, n_fft, hop):
    """The statistics as an epilogue of the forward kernel against the same statistics of the
    materialised magnitude spectrogram (same magnitudes; only the summation order differs)."""
    rng = np.random.default_rng(n_fft + hop)
    t = np.arange(30000) / 22050.0
    y = np.stack([np.sin(2 * np.pi * 900 * t) + 0.2 * rng.standard_normal(30000),
                  rng.standard_normal(30000), 0.01 * rng.standard_normal(30000)]).astype(np.float32)
    yd = _dev(y, gpu)
    S = magnitude(stft(yd, n_fft=n_fft, hop_length=hop))
    kw = dict(sr=22050, n_fft=n_fft)
    for p_exp, norm, roll, power in ((2.0, True, 0.85, 2.0), (1.0, False, 0.5, 1.0), (3.0, True, 0.99, 2.0)):
        got = fused_statistics(yd, hop_length=hop, p=p_exp, norm=norm, roll_percent=roll, power=power, **kw)
        assert got.shape == (3, 4, S.shape[-1])
        c = spectral_centroid(S=S, **kw)
        torch.testing.assert_close(got[:, 0:1], c, rtol=3e-5, atol=1e-2)
        torch.testing.assert_close(got[:, 1:2], spectral_bandwidth(S=S, p=p_exp, norm=norm, **kw), rtol=1e-4, atol=1e-2)
        r = spectral_rolloff(S=S, roll_percent=roll, **kw)
        step = 22050 / n_fft
        assert (got[:, 2:3] - r).abs().max().item() <= step * 1.001
        assert ((got[:, 2:3] != r).float().mean().item()) <= 0.03
        fl = spectral_flatness(S=S ** power if power != 1.0 else S)
        torch.testing.assert_close(got[:, 3:4], fl, rtol=2e-4, atol=1e-9)
    one = fused_statistics(yd[0], hop_length=hop, **kw)
    assert one.shape == (4, S.shape[-1])
    torch.testing.assert_close(one, fused_statistics(yd, hop_length=hop, **kw)[0], rtol=0, atol=0)


def test_all_four_in_one_launch_equal_the_single_statistics(gpu):
    from mlx_audio_amd.primitives import spectrogram_statistics

    rng = np.random.default_rng(21)
    for shape in ((3, 1025, 60), (513, 33), (2, 129, 300)):
        S = _dev(_spec(rng, shape), gpu)
        n_fft = 2 * (shape[-2] - 1)
        for p_exp, norm, roll in ((2.0, True, 0.85), (1.0, False, 0.3), (3.0, True, 1.0)):
            allf = spectrogram_statistics(S, sr=22050, n_fft=n_fft, p=p_exp, norm=norm, roll_percent=roll, power=2.0)
            kw = dict(S=S, sr=22050, n_fft=n_fft)
            assert torch.equal(allf[..., 0:1, :], spectral_centroid(**kw))
            assert torch.equal(allf[..., 1:2, :], spectral_bandwidth(p=p_exp, norm=norm, **kw))
            assert torch.equal(allf[..., 2:3, :], spectral_rolloff(roll_percent=roll, **kw))
            assert torch.equal(allf[..., 3:4, :], spectral_flatness(S=S * S))


def test_rolloff_checkpoint_scan_equals_the_full_scan(gpu):
    """The roll-off kernel restarts its second pass from a checkpoint of the running sum; the
    all-statistics kernel scans every bin again.  Same bits, whatever the columns look like:
    peaks at scattered bins (every lane crosses in another block), empty columns, infinities,
    and - through the fallback of a wave that meets them - negative values and NaNs."""
    from mlx_audio_amd.primitives import spectrogram_statistics

    rng = np.random.default_rng(77)
    for bins, frames in ((1025, 300), (2049, 70), (129, 1000), (33, 64), (7, 500), (4097, 65)):
        S = (rng.standard_normal((2, bins, frames)).astype(np.float32) ** 2) * 1e-3
        peak = rng.integers(0, bins, size=(2, frames))
        for r in range(2):
            S[r, peak[r], np.arange(frames)] += 50.0
        S[0, :, 3] = 0.0
        S[1, bins // 2, 5] = np.inf
        cases = {"plain": S}
        neg = S.copy()
        neg[0, 1:bins:3, 10:20] *= -1.0
        cases["negative"] = neg
        nan = S.copy()
        nan[1, bins - 1, 7] = np.nan
        cases["nan"] = nan
        n_fft = 2 * (bins - 1)
        for name, arr in cases.items():
            Sd = _dev(arr, gpu)
            for roll in (0.85, 0.3, 0.0, 1.0):
                r = spectral_rolloff(S=Sd, sr=22050, n_fft=n_fft, roll_percent=roll)
                allf = spectrogram_statistics(Sd, sr=22050, n_fft=n_fft, roll_percent=roll, power=2.0)
                assert torch.equal(allf[..., 2:3, :], r), (bins, name, roll)
                want = orc.spectral_rolloff(S=arr, sr=22050, n_fft=n_fft, roll_percent=roll)
                got = r.cpu().numpy()
                assert np.abs(got - want).max() <= 22050 / n_fft * 1.001, (bins, name, roll)
                assert (got != want).mean() <= 0.02, (bins, name, roll)


def test_short_signals_and_c_abi_checks(gpu):
    with pytest.raises(ValueError, match="must be >= frame_length"):
        zero_crossing_rate(_dev(np.zeros(10, np.float32), gpu), frame_length=64, center=False)
    lib = _lib.lib()
    S = torch.ones((4, 3), device=gpu)
    out = torch.zeros(3, device=gpu)
    f = torch.arange(4, device=gpu, dtype=torch.float32)
    s = torch.cuda.current_stream(gpu).cuda_stream
    assert lib.mia_spectral_stat(S.data_ptr(), 1, 4, 3, f.data_ptr(), 0, 7, 2.0, 1, 0.0, 0.0, 1.0,
                                 out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_spectral_stat(S.data_ptr(), 1, 4, 3, 0, 0, 0, 2.0, 1, 0.0, 0.0, 1.0,
                                 out.data_ptr(), s) == _lib.MIA_ERR_INVALID   # centroid without freq
    assert lib.mia_spectral_stat(S.data_ptr(), 1, 4, 3, f.data_ptr(), 0, 1, 2.0, 1, 0.0, 0.0, 1.0,
                                 out.data_ptr(), s) == _lib.MIA_ERR_INVALID   # bandwidth without centroid
    assert lib.mia_spectral_stat(S.data_ptr(), 0, 4, 3, f.data_ptr(), 0, 0, 2.0, 1, 0.0, 0.0, 1.0,
                                 out.data_ptr(), s) == _lib.MIA_OK
    assert lib.mia_spectral_stat(S.data_ptr(), 1, 4, 3, f.data_ptr(), 0, 0, 2.0, 1, 0.0, 0.0, 1.0,
                                 out.data_ptr(), s) == _lib.MIA_OK
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), 1.5, rtol=1e-6)
    assert lib.mia_zero_crossing_rate(S.data_ptr(), 1, 12, 0, 1, 1, 1, 0, out.data_ptr(),
                                      s) == _lib.MIA_ERR_INVALID


def test_full_size_batch_properties(gpu):
    """1024 x 30 s @16 kHz worth of frames (n_fft 2048, hop 512 -> 938 frames) as a power
    spectrogram: spot rows against the oracle; scaling laws on all rows (centroid, roll-off and
    flatness do not change when the spectrogram is scaled, bandwidth neither)."""
    g = torch.Generator(device=gpu)
    g.manual_seed(3)
    S = torch.rand((256, 1025, 938), generator=g, device=gpu) + 0.05
    c = spectral_centroid(S=S, sr=16000)
    b = spectral_bandwidth(S=S, sr=16000)
    r = spectral_rolloff(S=S, sr=16000)
    fl = spectral_flatness(S=S)
    rows = [0, 100, 255]
    Sc = S[rows].cpu().numpy()
    np.testing.assert_allclose(c[rows].cpu().numpy(), orc.spectral_centroid(S=Sc, sr=16000), rtol=RTOL)
    np.testing.assert_allclose(b[rows].cpu().numpy(), orc.spectral_bandwidth(S=Sc, sr=16000), rtol=5e-5)
    np.testing.assert_allclose(fl[rows].cpu().numpy(), orc.spectral_flatness(S=Sc), rtol=5e-5)
    assert np.abs(r[rows].cpu().numpy() - orc.spectral_rolloff(S=Sc, sr=16000)).max() <= 8000 / 1024 * 1.001
    S4 = S * 4.0   # exact in float32
    assert torch.allclose(spectral_centroid(S=S4, sr=16000), c, rtol=1e-6)
    assert torch.equal(spectral_rolloff(S=S4, sr=16000), r)
    assert torch.allclose(spectral_flatness(S=S4), fl, rtol=1e-5)
    assert torch.allclose(spectral_bandwidth(S=S4, sr=16000), b, rtol=1e-6)


def test_contrast_wide_quantile_bands():
    """More than 64 values per quantile in a band (44.1 kHz, quantile 0.1: the top band holds
    ~730 bins): those bands come from a device sort instead of the register arrays; against
    the oracle like every other case."""
    g = torch.Generator(device="cuda").manual_seed(5)
    S = torch.rand((3, 1025, 40), device="cuda", generator=g) + 0.01
    got = spectral_contrast(S=S, sr=44100, quantile=0.1).cpu().numpy()
    want = orc.spectral_contrast(S=S.cpu().numpy(), sr=44100, n_fft=2048, quantile=0.1)
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=2e-5, atol=2e-5)


def test_full_size_zero_crossings_and_db_round_trip(gpu):
    """1024 x 30 s @16 kHz through the span kernel: spot rows against the oracle bit for bit, and
    the size-independent property that a positive scale changes no count; 492 M decibel values
    back to powers and to decibels again."""
    from mlx_audio_amd.primitives import db_to_power, power_to_db

    g = torch.Generator(device=gpu)
    g.manual_seed(11)
    y = torch.randn((1024, 480000), generator=g, device=gpu)
    z = zero_crossing_rate(y)
    assert z.shape == (1024, 1, 938)
    rows = [0, 511, 1023]
    np.testing.assert_array_equal(z[rows].cpu().numpy(), orc.zero_crossing_rate(y[rows].cpu().numpy()))
    assert torch.equal(zero_crossing_rate(y * 3.0), z)
    z400 = zero_crossing_rate(y, frame_length=400, hop_length=160)
    np.testing.assert_array_equal(z400[rows].cpu().numpy(),
                                  orc.zero_crossing_rate(y[rows].cpu().numpy(), 400, 160))
    del z, z400
    S = torch.rand((512, 1025, 938), generator=g, device=gpu) * 100.0 + 1e-6
    db = power_to_db(S, top_db=None)
    back = db_to_power(db)
    assert torch.allclose(back, S, rtol=2e-5, atol=0.0)
    assert (power_to_db(back, top_db=None) - db).abs().max().item() <= 2e-5
