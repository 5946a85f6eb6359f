"""GPU parity: melspectrogram / Whisper log-mel / dB / DCT / MFCC and the
model-side front-ends against the CPU oracle.

Stated float32 tolerances
  * Whisper log-mel, (log10+4)/4 domain: max abs error <= 1e-4 (white noise and
    speech-like inputs); tonal inputs are checked on the bins above the
    float32 FFT noise floor (see test_logmel_tonal).
  * mel power: rtol 1e-4 (+ atol 1e-5 relative to the largest value), the
    reference's own mel tolerance (tests/parity/test_stft_parity.py:240-242).
  * dB features: 1e-3 dB.  MFCC: 2e-3 (the reference's cross-language MFCC
    tolerance is 1e-3 on smaller inputs).
"""
import numpy as np
import pytest
import torch

import oracle.spectral as o

pytestmark = pytest.mark.gpu


def _noise(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


def _speechlike(n, seed):
    """Amplitude-modulated coloured noise with silent gaps."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n).astype(np.float32)
    x = np.convolve(x, np.ones(8, np.float32) / 8, mode="same")
    env = (np.sin(np.linspace(0, 40, n)) > 0.2).astype(np.float32)
    env *= np.abs(np.sin(np.linspace(0, 7, n))).astype(np.float32)
    return (x * env).astype(np.float32)


@pytest.mark.parametrize("cfg", [
    dict(sr=22050, n_fft=2048, hop_length=512, n_mels=128),
    dict(sr=16000, n_fft=400, hop_length=160, n_mels=80, fmin=0.0, fmax=8000.0),
    dict(sr=16000, n_fft=400, hop_length=160, n_mels=128, fmin=0.0, fmax=8000.0),
    dict(sr=48000, n_fft=1024, hop_length=480, n_mels=64, fmin=50, fmax=14000, htk=True),
    dict(sr=44100, n_fft=4096, hop_length=1024, n_mels=128),
    dict(sr=8000, n_fft=256, hop_length=64, n_mels=40),
    dict(sr=16000, n_fft=512, hop_length=128, n_mels=64, norm=None, power=1.0),
    dict(sr=16000, n_fft=800, hop_length=200, n_mels=80, power=1.5, pad_mode="reflect"),
])
def test_melspectrogram_matches_oracle(gpu, cfg):
    from mlx_audio_amd.primitives import melspectrogram

    y = _noise((3, 30011), seed=cfg["n_fft"])
    got = melspectrogram(torch.from_numpy(y).to(gpu), **cfg).cpu().numpy()
    ref = o.melspectrogram(y, **cfg)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5 * ref.max())
    got1 = melspectrogram(torch.from_numpy(y[0]).to(gpu), **cfg).cpu().numpy()
    assert got1.shape == ref.shape[1:]
    np.testing.assert_allclose(got1, ref[0], rtol=1e-4, atol=1e-5 * ref.max())


@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_whisper_noise(gpu, n_mels):
    from mlx_audio_amd.frontends import compute_log_mel_spectrogram

    y = _noise((4, 480000), seed=1234)
    yt = torch.from_numpy(y).to(gpu)
    got = compute_log_mel_spectrogram(yt, n_mels=n_mels, per_clip=True).cpu().numpy()
    assert got.shape == (4, n_mels, 3001)
    for b in range(4):  # the reference's callers pass one clip at a time
        ref = o.compute_log_mel_spectrogram(y[b], n_mels=n_mels)
        assert np.abs(got[b] - ref).max() <= 1e-4
    # default = the reference's array-global maximum
    got_g = compute_log_mel_spectrogram(yt, n_mels=n_mels).cpu().numpy()
    assert np.abs(got_g - o.compute_log_mel_spectrogram(y, n_mels=n_mels)).max() <= 1e-4
    one = compute_log_mel_spectrogram(yt[0], n_mels=n_mels).cpu().numpy()
    assert one.shape == (n_mels, 3001)
    assert np.abs(one - o.compute_log_mel_spectrogram(y[0], n_mels=n_mels)).max() <= 1e-4


def test_logmel_whisper_clamped_inputs(gpu):
    """Inputs whose dynamic range exceeds 8 decades exercise the floor pass."""
    from mlx_audio_amd.frontends import compute_log_mel_spectrogram

    clips = np.stack([
        _speechlike(160000, 1),
        np.zeros(160000, np.float32),                                # silence -> -1.5
        _noise(160000, 3) * 1e-6,                                    # 1e-6-scaled
        np.concatenate([np.zeros(80000, np.float32), _noise(80000, 4)]),
    ])
    got = compute_log_mel_spectrogram(torch.from_numpy(clips).to(gpu), n_mels=128,
                                      per_clip=True).cpu().numpy()
    for b in range(len(clips)):
        ref = o.compute_log_mel_spectrogram(clips[b], n_mels=128)
        assert np.abs(got[b] - ref).max() <= 1e-4, b
    # log10(float32(1e-10)) is -10 only to within an ulp (NumPy gives -1.5000002 here)
    assert np.abs(got[1] + 1.5).max() <= 1e-6
    got_g = compute_log_mel_spectrogram(torch.from_numpy(clips).to(gpu), n_mels=128)
    ref_g = o.compute_log_mel_spectrogram(clips, n_mels=128)
    assert np.abs(got_g.cpu().numpy() - ref_g).max() <= 1e-4


def test_logmel_tonal(gpu):
    """Three sines + noise at -40 dB.  Bins far below the peak sit on the
    float32 FFT noise floor, where any two float32 FFTs (pocketfft included)
    disagree; there both are compared with the float64 truth instead."""
    from mlx_audio_amd.frontends import compute_log_mel_spectrogram

    t = np.arange(480000) / 16000.0
    y = (np.sin(2 * np.pi * 440 * t) + 0.5 * np.sin(2 * np.pi * 1000 * t) +
         0.25 * np.sin(2 * np.pi * 3000 * t)).astype(np.float32)
    y += 0.01 * _noise(480000, 9)
    got = compute_log_mel_spectrogram(torch.from_numpy(y).to(gpu), n_mels=128).cpu().numpy()
    ref32 = o.compute_log_mel_spectrogram(y, n_mels=128)
    ref64 = o.compute_log_mel_spectrogram(y, n_mels=128, dtype=np.float64)
    assert np.abs(got - ref32).max() <= 1e-4
    assert np.abs(got - ref64).max() <= max(1e-4, 2 * np.abs(ref32 - ref64).max())


def test_power_to_db_and_friends(gpu):
    from mlx_audio_amd.primitives import (amplitude_to_db, db_to_amplitude, db_to_power,
                                          power_to_db)

    p = torch.tensor([1.0, 10.0, 100.0], device=gpu)
    np.testing.assert_allclose(power_to_db(p).cpu().numpy(), [0, 10, 20], atol=1e-5)
    np.testing.assert_allclose(amplitude_to_db(p, amin=1e-5).cpu().numpy(), [0, 20, 40],
                               atol=1e-5)
    clip = power_to_db(torch.tensor([1e-20, 1.0, 100.0], device=gpu)).cpu().numpy()
    np.testing.assert_allclose(clip, [-60, 0, 20], atol=1e-5)
    S = np.abs(_noise((3, 64, 200), 5)) ** 2 * np.array([1e-6, 1.0, 1e3], np.float32)[:, None, None]
    St = torch.from_numpy(S).to(gpu)
    np.testing.assert_allclose(power_to_db(St).cpu().numpy(), o.power_to_db(S), atol=1e-4)
    np.testing.assert_allclose(power_to_db(St, top_db=None).cpu().numpy(),
                               o.power_to_db(S, top_db=None), atol=1e-4)
    per = power_to_db(St, per_clip=True).cpu().numpy()
    for b in range(3):
        np.testing.assert_allclose(per[b], o.power_to_db(S[b]), atol=1e-4)
    np.testing.assert_allclose(power_to_db(St, ref=torch.max).cpu().numpy(),
                               o.power_to_db(S, ref=np.max), atol=1e-4)
    np.testing.assert_allclose(amplitude_to_db(St).cpu().numpy(), o.amplitude_to_db(S), atol=1e-4)
    with pytest.raises(ValueError, match="top_db must be positive"):
        power_to_db(St, top_db=0)
    db = power_to_db(St, top_db=None)
    np.testing.assert_allclose(db_to_power(db).cpu().numpy(), np.maximum(S, 1e-10), rtol=1e-4)
    np.testing.assert_allclose(db_to_amplitude(amplitude_to_db(St, top_db=None)).cpu().numpy(),
                               np.maximum(S, 1e-5), rtol=1e-4)


@pytest.mark.parametrize("shape", [(7,), (3, 64, 201), (5, 4099), (0, 12), (1, 1)])
def test_db_to_power_and_amplitude_against_the_oracle(gpu, shape):
    """One pass of ref * 10^(x / 10 | 20) (convert.py:97-126, 166-195): float32 round-off of the
    oracle's power function (4e-6 relative: an exponent of up to 12 magnifies the last bit of the
    quotient), non-finite values as IEEE has them, a view that is not contiguous, an odd element count (the float4 path's tail)."""
    from mlx_audio_amd.primitives import db_to_amplitude, db_to_power

    rng = np.random.default_rng(len(shape) + int(np.prod(shape)))
    x = (rng.uniform(-120.0, 60.0, shape)).astype(np.float32)
    if x.size >= 7:
        x.reshape(-1)[:5] = [np.inf, -np.inf, np.nan, 0.0, -0.0]
    xd = torch.from_numpy(x).to(gpu)
    for fn, ofn in ((db_to_power, o.db_to_power), (db_to_amplitude, o.db_to_amplitude)):
        for ref in (1.0, 0.25, 3.0):
            got = fn(xd, ref=ref).cpu().numpy()
            want = ofn(x, ref=ref)
            assert got.shape == want.shape and got.dtype == np.float32
            np.testing.assert_allclose(got, want, rtol=4e-6, atol=0, equal_nan=True)
    if x.ndim == 3:
        view = xd.transpose(1, 2)[:, 1:, :-1]
        np.testing.assert_allclose(db_to_power(view).cpu().numpy(), o.db_to_power(view.cpu().numpy()),
                                   rtol=4e-6, equal_nan=True)
        np.testing.assert_allclose(db_to_power(x).cpu().numpy(), o.db_to_power(x), rtol=4e-6, equal_nan=True)


def test_dct(gpu):
    import scipy.fft

    from mlx_audio_amd.primitives import dct

    x = _noise((3, 128, 77), 6)
    xt = torch.from_numpy(x).to(gpu)
    got = dct(xt, n=40, axis=1).cpu().numpy()
    np.testing.assert_allclose(got, o.dct(x, n=40, axis=1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got, scipy.fft.dct(x, type=2, axis=1, norm="ortho")[:, :40],
                               rtol=1e-4, atol=1e-4)
    assert dct(xt, n=20).shape == (3, 128, 20)
    np.testing.assert_allclose(dct(xt, axis=-1).cpu().numpy(), o.dct(x), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(dct(xt, axis=0, norm=None).cpu().numpy(),
                               o.dct(x, axis=0, norm=None), rtol=1e-5, atol=2e-5)
    with pytest.raises(ValueError, match="Only DCT type 2"):
        dct(xt, type=3)


@pytest.mark.parametrize("cfg", [
    dict(sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128),
    dict(sr=22050, n_mfcc=13, lifter=22),
    dict(sr=16000, n_mfcc=20, n_fft=400, hop_length=160, n_mels=80),
])
def test_mfcc_matches_oracle(gpu, cfg):
    from mlx_audio_amd.primitives import mfcc

    y = np.stack([_noise(48000, 7), _speechlike(48000, 8) * 3.0])
    yt = torch.from_numpy(y).to(gpu)
    got = mfcc(yt, **cfg).cpu().numpy()
    ref = o.mfcc(y, **cfg)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-3)
    per = mfcc(yt, per_clip=True, **cfg).cpu().numpy()
    for b in range(2):
        np.testing.assert_allclose(per[b], o.mfcc(y[b], **cfg), rtol=1e-5, atol=2e-3)
    # pre-computed log-mel path
    S = o.power_to_db(o.melspectrogram(y[0], sr=16000, n_fft=512, n_mels=40))
    np.testing.assert_allclose(mfcc(S=torch.from_numpy(S).to(gpu), n_mfcc=13).cpu().numpy(),
                               o.mfcc(S=S, n_mfcc=13), rtol=0, atol=1e-3)
    with pytest.raises(ValueError, match="n_mfcc must be positive"):
        mfcc(yt, n_mfcc=0)


def test_htdemucs_spec_ispec(gpu):
    from mlx_audio_amd.frontends import htdemucs_ispec, htdemucs_spec

    x = (_noise((1, 2, 44100 * 3 + 17), 42) * 0.1).astype(np.float32)
    got = htdemucs_spec(torch.from_numpy(x).to(gpu))
    ref = o.htdemucs_spec(x)
    assert tuple(got.shape) == ref.shape == (1, 2, 2048, 130)
    assert np.abs(got.cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max()
    assert np.abs(got.cpu().numpy() - ref).mean() < 1e-3  # the reference's MAE bound
    # inverse: build the [B, S, 2C, F, T] real layout from the spectrum, S=1
    B, C, F, Tf = ref.shape
    m = np.stack([ref.real, ref.imag], axis=2).reshape(B, 1, 2 * C, F, Tf).astype(np.float32)
    rec = htdemucs_ispec(torch.from_numpy(m).to(gpu), length=x.shape[-1]).cpu().numpy()
    ref_rec = o.htdemucs_ispec(m, length=x.shape[-1])
    assert rec.shape == ref_rec.shape == (1, 1, 2, x.shape[-1])
    assert np.abs(rec - ref_rec).max() <= 1e-5
    # away from the clip ends the chain reproduces the input (the dropped
    # Nyquist bin costs a little energy, hence not exact)
    assert np.abs(rec[0, 0, :, 8192:-8192] - x[0, :, 8192:-8192]).max() < 5e-3


def test_clap_features(gpu):
    from mlx_audio_amd.frontends import CLAPFeatureExtractor, clap_log_mel

    y = _noise((2, 480000), 7)
    got = clap_log_mel(torch.from_numpy(y).to(gpu)).cpu().numpy()
    ref = o.clap_mel(y)
    assert got.shape == ref.shape == (2, 1001, 64)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-3)
    fe = CLAPFeatureExtractor(rng=np.random.default_rng(0))
    short = _noise(100000, 8)
    long = _noise(480000 + 96000, 9)
    out = fe([torch.from_numpy(short), torch.from_numpy(long)])
    assert tuple(out["input_features"].shape) == (2, 4, 1001, 64)
    assert out["is_longer"].cpu().numpy().tolist() == [[False], [True]]
    np.testing.assert_allclose(out["input_features"][0].cpu().numpy(), o.clap_padded_mel(short),
                               rtol=0, atol=1e-3)
    full = o.clap_mel(long)
    crops = out["input_features"][1].cpu().numpy()
    for c in range(1, 4):  # each crop is a contiguous window of the full mel
        d = [np.abs(full[s:s + 1001] - crops[c]).max() for s in range(full.shape[0] - 1000)]
        assert min(d) <= 1e-3


@pytest.mark.parametrize("extra", [96000, 480000 * 2 + 777, 481])
def test_clap_fusion_stack_against_oracle(gpu, extra):
    """All four channels of the fusion stack of a long clip (feature_extractor.py:233-277)
    against the oracle's restatement with the same generator: the bilinear shrink of the full
    mel (channel 0) and the three crops at the same start frames."""
    from mlx_audio_amd.frontends import CLAPFeatureExtractor

    long = _noise(480000 + extra, 20 + extra % 7)
    fe = CLAPFeatureExtractor(rng=np.random.default_rng(5))
    got = fe([torch.from_numpy(long).to(gpu)])["input_features"][0].cpu().numpy()
    want, starts = o.clap_fusion_mel(long, np.random.default_rng(5))
    assert got.shape == want.shape == (4, 1001, 64)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-3)  # dB values; the tolerance of the dB tests


@pytest.mark.parametrize("T", [400, 401, 1000, 10239, 10240, 10241, 16001, 20479, 20483])
def test_logmel_whisper_awkward_lengths(gpu, T):
    """Clip lengths around the 64-frame tile size, odd lengths (clips after the first are then
    not 16-byte aligned, which takes the per-sample staging path) and clips shorter than a tile."""
    from mlx_audio_amd.frontends import compute_log_mel_spectrogram

    rng = np.random.default_rng(T)
    x = rng.standard_normal((3, T)).astype(np.float32)
    got = compute_log_mel_spectrogram(torch.from_numpy(x).to(gpu), n_mels=80, per_clip=True)
    assert tuple(got.shape) == (3, 80, 1 + T // 160)
    got = got.cpu().numpy()
    for b in range(3):
        ref = o.compute_log_mel_spectrogram(x[b], n_mels=80)
        assert np.abs(got[b] - ref).max() <= 1e-4, (T, b)


def test_logmel_whisper_unswept_filterbank_uses_generic_kernel(gpu):
    """A filterbank that is not of the two-taps-per-bin triangular form (here: power 1.0, and a
    filterbank with fmax below Nyquist and HTK spacing) must still match the oracle."""
    from mlx_audio_amd.primitives import melspectrogram

    rng = np.random.default_rng(77)
    x = rng.standard_normal((2, 16000)).astype(np.float32)
    for kw in (dict(power=1.0), dict(htk=True, fmin=100.0, fmax=6000.0, n_mels=40),
               dict(norm=None, n_mels=20)):
        got = melspectrogram(torch.from_numpy(x).to(gpu), sr=16000, n_fft=400, hop_length=160, **kw)
        ref = o.melspectrogram(x, sr=16000, n_fft=400, hop_length=160, **kw)
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * ref.max())


def test_power_to_db_callable_ref():
    """`ref=np.max` (the usual idiom) works on a device tensor, as does a torch reducer."""
    from mlx_audio_amd.primitives import power_to_db

    g = torch.Generator(device="cuda").manual_seed(2)
    S = torch.rand((4, 64, 50), device="cuda", generator=g) + 1e-3
    a = power_to_db(S, ref=np.max).cpu().numpy()
    b = power_to_db(S, ref=torch.max).cpu().numpy()
    c = power_to_db(S, ref=float(S.max())).cpu().numpy()
    assert np.array_equal(a, c) and np.array_equal(b, c)
    assert abs(a.max()) < 1e-6


@pytest.mark.parametrize("n_fft,hop,n_mels", [(1024, 480, 64), (2048, 512, 128), (512, 128, 40), (400, 100, 80),
                                              (4096, 1024, 96), (256, 64, 32), (800, 200, 60)])
def test_frame_major_mel_equals_transposed_mel(gpu, n_fft, hop, n_mels):
    """mia_melspectrogram_frames writes (B, F, n_mels); the values are those of
    mia_melspectrogram bit for bit, for the dB epilogue with a clamp and for plain mel power.
    Plans it does not serve (the n_fft = 400 / hop 160 kernel, n_mels % 4 != 0) say so."""
    from mlx_audio_amd import _lib
    from mlx_audio_amd.primitives.mel import _mel_raw

    g = torch.Generator(device="cuda").manual_seed(n_fft + n_mels)
    y = torch.randn((3, 5 * n_fft + 37), device=gpu, generator=g)
    y[1, : 2 * n_fft] *= 1e-3
    for kw in ({"epilogue": _lib.EPI_DB, "top_db": 60.0, "per_clip": True}, {}):
        a = _mel_raw(y, 16000, n_fft, hop, None, "hann", True, "reflect", 2.0, n_mels, 0.0, None, False,
                     "slaney", **kw)
        b = _mel_raw(y, 16000, n_fft, hop, None, "hann", True, "reflect", 2.0, n_mels, 0.0, None, False,
                     "slaney", frames_first=True, **kw)
        assert b.shape == (3, a.shape[2], n_mels) and b.is_contiguous()
        assert torch.equal(a.transpose(1, 2), b)


def test_frame_major_mel_unsupported_plans(gpu):
    from mlx_audio_amd import _lib
    from mlx_audio_amd.primitives.mel import _mel_raw

    y = torch.randn((2, 16000), device=gpu)
    # Whisper's framing is served by its own kernel: the mirror falls back to a transposed view
    a = _mel_raw(y, 16000, 400, 160, None, "hann", True, "constant", 2.0, 80, 0.0, 8000.0, False, "slaney")
    b = _mel_raw(y, 16000, 400, 160, None, "hann", True, "constant", 2.0, 80, 0.0, 8000.0, False, "slaney",
                 frames_first=True)
    assert not b.is_contiguous() and torch.equal(a.transpose(1, 2), b)
    c = _mel_raw(y, 16000, 1024, 256, None, "hann", True, "constant", 2.0, 30, 0.0, None, False, "slaney",
                 frames_first=True)
    assert c.shape[2] == 30 and not c.is_contiguous()


@pytest.mark.parametrize("hop,n_mels,pad_mode,served", [
    (480, 64, "reflect", True), (256, 128, "constant", True), (384, 80, "edge", True), (170, 32, "reflect", True),
    (512, 80, "edge", False),      # four frames' samples per wave no longer fit the LDS beside the exchange rows
    (480, 40, "reflect", False)])  # bands in sixteens only
def test_wave_private_kernel_option(gpu, hop, n_mels, pad_mode, served):
    """Plan option "fwd_lane": frame-major mel rows at n_fft 1024 from the kernel in which four
    frames share a wave (16 lanes each, radix 32 x 16 with one exchange) instead of the four-pass
    kernel.  Another order of additions, so float32 round-off apart from the default path; both
    against the float64 oracle as well.  Edge frames, a ragged frame count and a quiet stretch;
    plans outside the kernel's limits keep the default path."""
    from mlx_audio_amd import _lib
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives.mel import _mel_raw, mel_filterbank_np, padded_window_np

    g = torch.Generator(device="cuda").manual_seed(hop + n_mels)
    y = torch.randn((3, 7 * 1024 + 333), device=gpu, generator=g)
    y[1, :3000] *= 1e-3
    args = (y, 16000, 1024, hop, None, "hann", True, pad_mode, 2.0, n_mels, 0.0, None, False, "slaney")
    plan = get_plan(y.device, 1024, hop, True, pad_mode, padded_window_np("hann", 1024, 1024),
                    mel_basis=mel_filterbank_np(16000, 1024, n_mels, 0.0, None, False, "slaney"), power=2.0)
    lib = _lib.lib()
    for kw in ({}, {"epilogue": _lib.EPI_DB, "top_db": 80.0, "per_clip": True}):
        want = _mel_raw(*args, frames_first=True, **kw)
        try:
            _lib.check(lib.mia_plan_set_option(plan.handle, b"fwd_lane", 1))
            got = _mel_raw(*args, frames_first=True, **kw)
        finally:
            lib.mia_plan_set_option(plan.handle, b"fwd_lane", 0)
        assert got.shape == want.shape and got.is_contiguous()
        if not served:  # plans the kernel does not serve stay on the four-pass kernel
            assert torch.equal(got, want)
            continue
        assert not torch.equal(got, want)  # the option did switch kernels
        if kw:
            assert (got - want).abs().max().item() <= 2e-4  # dB
        else:
            ref = o.melspectrogram(y.cpu().numpy(), sr=16000, n_fft=1024, hop_length=hop, n_mels=n_mels,
                                   pad_mode=pad_mode).transpose(0, 2, 1)
            np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * ref.max())
            np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=1e-6 * ref.max())


@pytest.mark.parametrize("n_fft,hop,n_mels,sr", [(480, 160, 80, 16000), (960, 480, 64, 48000), (882, 441, 40, 44100),
                                                (1200, 300, 128, 48000), (320, 80, 24, 16000), (64, 16, 20, 16000),
                                                (128, 32, 40, 16000)])
def test_mel_chain_on_the_per_wave_transform(gpu, n_fft, hop, n_mels, sr):
    """n_fft without a compiled plan but with a 7-smooth half length (20 / 30 ms windows at 16,
    44.1 and 48 kHz): mel power, magnitude mel, dB with a per-clip top_db floor, frame-major rows and
    MFCC against the oracle; rows of a batch equal the rows computed alone."""
    from mlx_audio_amd import _lib
    from mlx_audio_amd.primitives import melspectrogram, mfcc, power_to_db
    from mlx_audio_amd.primitives.mel import _mel_raw

    assert _lib.lib().mia_fft_size_supported(n_fft) == 0
    g = torch.Generator(device="cuda").manual_seed(n_fft + n_mels)
    y = 0.3 * torch.randn((3, 9 * n_fft + 123), device=gpu, generator=g)
    y[1, : 4 * n_fft] *= 1e-4
    x = y.cpu().numpy()
    got = melspectrogram(y, sr=sr, n_fft=n_fft, hop_length=hop, n_mels=n_mels, pad_mode="reflect")
    ref = o.melspectrogram(x, sr=sr, n_fft=n_fft, hop_length=hop, n_mels=n_mels, pad_mode="reflect")
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * ref.max())
    one = melspectrogram(y[1], sr=sr, n_fft=n_fft, hop_length=hop, n_mels=n_mels, pad_mode="reflect")
    assert torch.equal(one, got[1])
    got1 = melspectrogram(y, sr=sr, n_fft=n_fft, hop_length=hop, n_mels=n_mels, power=1.0)
    ref1 = o.melspectrogram(x, sr=sr, n_fft=n_fft, hop_length=hop, n_mels=n_mels, power=1.0)
    np.testing.assert_allclose(got1.cpu().numpy(), ref1, rtol=1e-4, atol=1e-5 * ref1.max())
    # dB with the floor 60 dB below each clip's maximum, row-major and frame-major
    args = (y, sr, n_fft, hop, None, "hann", True, "constant", 2.0, n_mels, 0.0, None, False, "slaney")
    kw = {"epilogue": _lib.EPI_DB, "top_db": 60.0, "per_clip": True}
    a = _mel_raw(*args, **kw)
    want = np.stack([o.power_to_db(o.melspectrogram(x[i], sr=sr, n_fft=n_fft, hop_length=hop, n_mels=n_mels),
                                   top_db=60.0) for i in range(3)])
    assert np.abs(a.cpu().numpy() - want).max() <= 2e-3
    b = _mel_raw(*args, frames_first=True, **kw)
    assert b.shape == (3, a.shape[2], n_mels) and torch.equal(a.transpose(1, 2), b)
    c = mfcc(y, sr=sr, n_mfcc=13, n_fft=n_fft, hop_length=hop, n_mels=n_mels, per_clip=True)
    wantc = np.stack([o.mfcc(x[i], sr=sr, n_mfcc=13, n_fft=n_fft, hop_length=hop, n_mels=n_mels) for i in range(3)])
    np.testing.assert_allclose(c.cpu().numpy(), wantc, rtol=1e-4, atol=5e-3)


def test_power_to_db_shapes_and_floors(gpu):
    """The stand-alone conversion on arrays whose rows do / do not reach below the floor, with the
    maximum per clip and over everything, on odd row lengths (scalar path) and whole quads."""
    from mlx_audio_amd.primitives import amplitude_to_db, power_to_db

    rng = np.random.default_rng(3)
    for shape in ((5, 64, 101), (3, 128, 400), (1, 7), (4, 33, 1)):
        S = (rng.random(shape) + 1e-3).astype(np.float32)
        S[0].flat[::7] *= 1e-12     # this row reaches far below the floor
        if shape[0] > 2:
            S[2] *= 1e3             # ... and this one holds the global maximum
        x = torch.from_numpy(S).to(gpu)
        for top_db in (80.0, 20.0, None):
            got = power_to_db(x, top_db=top_db).cpu().numpy()
            np.testing.assert_allclose(got, o.power_to_db(S, top_db=top_db), rtol=0, atol=2e-4)
            if S.ndim >= 2:
                got = power_to_db(x, top_db=top_db, per_clip=True).cpu().numpy()
                want = np.stack([o.power_to_db(S[i], top_db=top_db) for i in range(S.shape[0])])
                np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)
        got = amplitude_to_db(x, ref=2.0, top_db=30.0).cpu().numpy()
        np.testing.assert_allclose(got, o.amplitude_to_db(S, ref=2.0, top_db=30.0), rtol=0, atol=2e-4)
