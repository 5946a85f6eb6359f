"""GPU edge cases: empty batches, clips shorter than a tile or than the padding,
non-finite samples (isolation between clips), overflowing power, and the plan
choices around the n_fft = 400 kernel.  Everything goes through the C ABI."""
import numpy as np
import pytest
import torch

import oracle.spectral as o

pytestmark = pytest.mark.gpu


def test_empty_batches(gpu):
    from mlx_audio_amd.frontends import compute_log_mel_spectrogram
    from mlx_audio_amd.primitives import istft, melspectrogram, mfcc, stft

    y = torch.zeros((0, 16000), device=gpu)
    assert tuple(stft(y, n_fft=512).shape) == (0, 257, 126)
    assert tuple(melspectrogram(y, sr=16000, n_fft=512, n_mels=40).shape) == (0, 40, 126)
    assert tuple(mfcc(y, sr=16000, n_fft=512, hop_length=128, n_mels=40, n_mfcc=13).shape) == (0, 13, 126)
    assert tuple(compute_log_mel_spectrogram(y, n_mels=80, per_clip=True).shape) == (0, 80, 101)
    S = torch.zeros((0, 257, 9), dtype=torch.complex64, device=gpu)
    assert tuple(istft(S, hop_length=128).shape) == (0, 8 * 128)


@pytest.mark.parametrize("T", [1, 2, 159, 160, 161, 201, 399, 400, 401, 640, 10239, 10240, 10241])
@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_very_short_clips(gpu, T, n_mels):
    """Clips from one sample up to just past one 64-frame tile: some frames consist of
    (zero) padding almost entirely."""
    from mlx_audio_amd.frontends import compute_log_mel_spectrogram

    rng = np.random.default_rng(T)
    y = rng.standard_normal((3, T)).astype(np.float32)
    got = compute_log_mel_spectrogram(torch.from_numpy(y).to(gpu), n_mels=n_mels, per_clip=True)
    got = got.cpu().numpy()
    for b in range(3):
        ref = o.compute_log_mel_spectrogram(y[b], n_mels=n_mels)
        assert got[b].shape == ref.shape
        assert np.abs(got[b] - ref).max() <= 1e-4, (T, b)


@pytest.mark.parametrize("T", [150, 200, 201, 4000])
def test_reflect_padding_needs_a_long_enough_signal(gpu, T):
    """Reflect padding of n_fft // 2 needs more than that many samples (the reference's
    slices come out short otherwise): ValueError below, oracle parity above."""
    from mlx_audio_amd.primitives import melspectrogram

    y = np.random.default_rng(T).standard_normal((2, T)).astype(np.float32)
    kw = dict(sr=16000, n_fft=400, hop_length=160, n_mels=80, fmin=0.0, fmax=8000.0, pad_mode="reflect")
    if T <= 200:
        with pytest.raises(ValueError):
            melspectrogram(torch.from_numpy(y).to(gpu), **kw)
        return
    got = melspectrogram(torch.from_numpy(y).to(gpu), **kw).cpu().numpy()
    ref = o.melspectrogram(y, **kw)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5 * ref.max())


def test_non_finite_samples_stay_in_their_clip(gpu):
    """A NaN or an infinity in one clip must not change any other clip of the batch
    (per-clip extremes live in registers across tiles and clips inside the kernel), and
    inside its own clip it behaves as in the reference: the frames that see the sample
    are NaN, the clip maximum is NaN, and maximum(x, NaN - 8) floods the whole clip."""
    from mlx_audio_amd.frontends import compute_log_mel_spectrogram

    gen = torch.Generator(device=gpu)
    gen.manual_seed(5)
    y = torch.randn((12, 48000), generator=gen, device=gpu)
    for n_mels in (80, 128):
        clean = compute_log_mel_spectrogram(y, n_mels=n_mels, per_clip=True)
        bad = y.clone()
        bad[3, 20000] = float("nan")
        bad[7, 100] = float("inf")
        bad[8, 47999] = -float("inf")
        out = compute_log_mel_spectrogram(bad, n_mels=n_mels, per_clip=True)
        for b in range(12):
            if b in (3, 7, 8):
                with np.errstate(invalid="ignore"):
                    ref = o.compute_log_mel_spectrogram(bad[b].cpu().numpy(), n_mels=n_mels)
                assert np.isnan(ref).all()
                assert torch.isnan(out[b]).all(), b
            else:
                assert torch.equal(out[b], clean[b]), b
        # array-wide maximum (the reference's batched call): one NaN floods everything
        assert torch.isnan(compute_log_mel_spectrogram(bad, n_mels=n_mels, per_clip=False)).all()


def test_nan_floods_db_and_mfcc_like_the_reference(gpu):
    """power_to_db(top_db=80) clamps against the maximum: a NaN maximum makes every value
    of the clip NaN in the reference, in the mel dB path, in mfcc and in the stand-alone op."""
    from mlx_audio_amd.primitives import mfcc, power_to_db

    rng = np.random.default_rng(9)
    y = rng.standard_normal((3, 30000)).astype(np.float32)
    y[1, 12345] = np.nan
    yt = torch.from_numpy(y).to(gpu)
    kw = dict(sr=16000, n_mfcc=13, n_fft=512, hop_length=128, n_mels=40)
    got = mfcc(yt, per_clip=True, **kw).cpu().numpy()
    for b in range(3):
        with np.errstate(invalid="ignore"):
            ref = o.mfcc(y[b], **kw)
        assert np.array_equal(np.isnan(got[b]), np.isnan(ref)), b
        if b != 1:
            np.testing.assert_allclose(got[b], ref, rtol=1e-5, atol=2e-3)
    S = np.abs(rng.standard_normal((2, 40, 50))).astype(np.float32) + 1e-3
    S[0, 3, 7] = np.nan
    got = power_to_db(torch.from_numpy(S).to(gpu)).cpu().numpy()
    with np.errstate(invalid="ignore"):
        ref = o.power_to_db(S)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.isnan(ref).all()  # the reference's maximum spans the whole array


def test_power_overflow_matches_oracle_where_finite(gpu):
    """|x| ~ 1e19 overflows float32 power to +inf: log-mel of those bins is +inf in the
    reference's arithmetic too; the rest of the batch is untouched."""
    from mlx_audio_amd.frontends import compute_log_mel_spectrogram

    rng = np.random.default_rng(11)
    y = rng.standard_normal((2, 16000)).astype(np.float32)
    y[1] *= 1e19
    got = compute_log_mel_spectrogram(torch.from_numpy(y).to(gpu), n_mels=80, per_clip=True).cpu().numpy()
    ref0 = o.compute_log_mel_spectrogram(y[0], n_mels=80)
    assert np.abs(got[0] - ref0).max() <= 1e-4
    with np.errstate(over="ignore", invalid="ignore"):
        ref1 = o.compute_log_mel_spectrogram(y[1], n_mels=80)
    assert np.array_equal(np.isfinite(got[1]), np.isfinite(ref1))


def test_dense_filterbank_takes_the_generic_kernel(gpu):
    """A filterbank that is not triangular (three non-zeros in a column) cannot use the
    sweep of the n_fft = 400 kernel; the plan falls back to the generic kernel and the
    result still matches the oracle's dense product."""
    from mlx_audio_amd._runtime import get_plan, stream_ptr
    from mlx_audio_amd import _lib
    from mlx_audio_amd.primitives.stft import padded_window_np

    rng = np.random.default_rng(3)
    basis = np.abs(rng.standard_normal((20, 201))).astype(np.float32)  # fully dense
    win = padded_window_np("hann", 400, 400)
    y = rng.standard_normal((2, 16000)).astype(np.float32)
    yt = torch.from_numpy(y).to(gpu)
    plan = get_plan(yt.device, 400, 160, True, "reflect", win, mel_basis=basis, power=2.0)
    lib = _lib.lib()
    F = lib.mia_plan_num_frames(plan.handle, 16000)
    out = torch.empty((2, 20, F), dtype=torch.float32, device=gpu)
    with torch.cuda.device(yt.device):
        _lib.check(lib.mia_melspectrogram(plan.handle, yt.data_ptr(), 2, 16000, 0, 1.0, 1e-10, 10.0,
                                          -1.0, 1, out.data_ptr(), None, 0, stream_ptr(yt.device)))
    S = np.abs(o.stft(y, n_fft=400, hop_length=160, pad_mode="reflect")) ** 2
    ref = np.einsum("mk,bkf->bmf", basis.astype(np.float64), S.astype(np.float64))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-4)
