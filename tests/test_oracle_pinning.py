"""Pins the CPU oracle (oracle/spectral.py) without the reference.

The reference's array library (MLX) is absent offline and its tests store no
numbers for this path, so the oracle is checked against
  * independent implementations that are importable here (CPU torch.stft /
    torch.istft, scipy.signal.get_window, scipy.fft.dct,
    transformers.audio_utils.mel_filter_bank),
  * the analytic known-answer cases of the reference's own test-suites
    (swift/Tests/MLXAudioPrimitivesTests/MelTests.swift:12-57,153-209,
    STFTTests.swift:12-92, MFCCTests.swift:12-32), and
  * the tolerances of python/tests/parity/test_stft_parity.py
    (STFT magnitude rtol 1e-4 / atol 1e-5 .. 1.5e-4, round trip rtol 1e-3 /
    atol 1e-4, filterbank rtol 1e-4 / atol 1e-6, mel rtol 1e-4 / atol 1e-5).
All tests here run on the CPU.
"""

import math

import numpy as np
import pytest
import scipy.fft
import scipy.signal
import torch

import oracle.spectral as o


def _torch_stft(x, n_fft, hop, center, pad_mode, win):
    xt = torch.from_numpy(np.asarray(x, dtype=np.float64))
    return torch.stft(xt, n_fft, hop_length=hop, win_length=n_fft,
                      window=torch.from_numpy(win.astype(np.float64)), center=center,
                      pad_mode=pad_mode, normalized=False, onesided=True,
                      return_complex=True).numpy()


# ----------------------------------------------------------------------------- windows


@pytest.mark.parametrize("name,scipy_name", [("hann", "hann"), ("hanning", "hann"),
                                              ("hamming", "hamming"), ("blackman", "blackman"),
                                              ("bartlett", "bartlett"), ("triangular", "bartlett"),
                                              ("rectangular", "boxcar"), ("boxcar", "boxcar"),
                                              ("ones", "boxcar")])
@pytest.mark.parametrize("n", [4, 255, 400, 1024])
@pytest.mark.parametrize("fftbins", [True, False])
def test_window_matches_scipy(name, scipy_name, n, fftbins):
    ours = o.get_window(name, n, fftbins)
    ref = scipy.signal.get_window(scipy_name, n, fftbins=fftbins)
    assert ours.dtype == np.float32 and ours.shape == (n,)
    np.testing.assert_allclose(ours, ref, rtol=0, atol=1e-7)
    assert ours.min() >= 0.0  # the reference clamps blackman's tiny negatives (windows.py:96-99)


def test_window_errors_and_array_window():
    with pytest.raises(ValueError):
        o.get_window("kaiser", 16)
    with pytest.raises(ValueError):
        o.get_window(np.ones(8, dtype=np.float32), 16)
    with pytest.raises(TypeError):
        o.get_window(3.0, 16)
    w = np.linspace(0, 1, 16, dtype=np.float32)
    np.testing.assert_array_equal(o.get_window(w, 16), w)


def test_padded_window_is_centred():
    w = o.padded_window("hann", 6, 10)
    assert w.shape == (10,)
    np.testing.assert_array_equal(w[:2], 0)
    np.testing.assert_array_equal(w[8:], 0)
    np.testing.assert_allclose(w[2:8], o.get_window("hann", 6), atol=0)


# ----------------------------------------------------------------------------- stft / istft


@pytest.mark.parametrize("n_fft,hop", [(256, 64), (400, 160), (512, 128), (800, 200),
                                       (1024, 480), (2048, 512), (4096, 1024)])
@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
def test_stft_matches_torch(n_fft, hop, pad_mode):
    rng = np.random.default_rng(123)
    x = rng.standard_normal((2, 16000)).astype(np.float32)
    win = o.get_window("hann", n_fft)
    ref = _torch_stft(x, n_fft, hop, True, pad_mode, win)
    got64 = o.stft(x, n_fft, hop, pad_mode=pad_mode, dtype=np.float64)
    got32 = o.stft(x, n_fft, hop, pad_mode=pad_mode)
    assert got32.dtype == np.complex64 and got32.shape == ref.shape
    np.testing.assert_allclose(got64, ref, rtol=0, atol=1e-9 * n_fft)
    # reference recipe (test_stft_parity.py:65-101): magnitudes, rtol 1e-4 / atol 1e-5,
    # atol scaled with the spectrum's size for the long transforms
    np.testing.assert_allclose(np.abs(got32), np.abs(ref), rtol=1e-4, atol=1e-5 * n_fft / 256)


def test_stft_uncentred_and_1d():
    rng = np.random.default_rng(42)
    x = rng.standard_normal(22050).astype(np.float32)
    win = o.get_window("hann", 2048)
    ref = _torch_stft(x, 2048, 512, False, "constant", win)
    got = o.stft(x, 2048, 512, center=False)
    assert got.shape == ref.shape == (1025, 1 + (22050 - 2048) // 512)
    np.testing.assert_allclose(np.abs(got), np.abs(ref), rtol=1e-4, atol=1e-4)


def test_stft_edge_padding_restated_directly():
    rng = np.random.default_rng(5)
    x = rng.standard_normal(3000)
    got = o.stft(x, 512, 128, pad_mode="edge", dtype=np.float64)
    xp = np.pad(x, 256, mode="edge")
    win = o.get_window("hann", 512).astype(np.float64)
    frames = np.lib.stride_tricks.sliding_window_view(xp, 512)[::128]
    np.testing.assert_allclose(got, np.fft.rfft(frames * win, axis=-1).T, atol=1e-10)


def test_stft_shapes_defaults_and_errors():
    # STFTTests.swift:12-47: (1025, > 40) for one second at 22050 Hz with the defaults
    S = o.stft(np.zeros(22050, dtype=np.float32))
    assert S.shape == (1025, 1 + 22050 // 512)
    for kw in (dict(hop_length=0), dict(win_length=0), dict(win_length=4096), dict(hop_length=4096)):
        with pytest.raises(ValueError):
            o.stft(np.zeros(22050, dtype=np.float32), **kw)
    with pytest.raises(ValueError):
        o.stft(np.zeros(22050, dtype=np.float32), pad_mode="wrap")
    # a 10 ms clip (441 samples) is shorter than the frame unless centre padding helps
    with pytest.raises(ValueError, match="must be >= frame_length"):
        o.stft(np.zeros(441, dtype=np.float32), n_fft=2048, center=False)
    assert o.stft(np.zeros(441, dtype=np.float32), n_fft=2048).shape == (1025, 1)


def test_sine_peaks_at_its_bin():
    sr, f0 = 22050, 440.0
    t = np.arange(sr) / sr
    S = np.abs(o.stft(np.sin(2 * np.pi * f0 * t).astype(np.float32), 2048, 512))
    assert int(np.argmax(S[:, 10])) == int(round(f0 * 2048 / sr))


@pytest.mark.parametrize("n_fft,hop", [(400, 160), (512, 128), (2048, 512), (4096, 1024)])
def test_round_trip(n_fft, hop):
    rng = np.random.default_rng(42)
    x = rng.standard_normal(22050).astype(np.float32)
    y = o.istft(o.stft(x, n_fft, hop), hop, length=len(x))
    np.testing.assert_allclose(y, x, rtol=1e-3, atol=1e-4)  # test_stft_parity.py:103-118
    snr = 10 * np.log10(np.sum(x.astype(np.float64) ** 2) /
                        np.sum((x.astype(np.float64) - y) ** 2))
    assert snr >= 100.0, snr


def test_istft_matches_torch():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 12000))
    win = o.get_window("hann", 512).astype(np.float64)
    S = o.stft(x, 512, 128, dtype=np.float64)
    ref = torch.istft(torch.from_numpy(S), 512, hop_length=128, window=torch.from_numpy(win),
                      center=True, length=12000).numpy()
    got = o.istft(S, 128, length=12000, dtype=np.float64)
    # torch divides by the raw window envelope, the reference by max(envelope, 1e-8):
    # identical wherever the envelope is not tiny, i.e. away from the first samples
    np.testing.assert_allclose(got[:, 8:-8], ref[:, 8:-8], atol=1e-9)
    got32 = o.istft(S.astype(np.complex64), 128, length=12000)
    assert got32.dtype == np.float32
    np.testing.assert_allclose(got32[:, 8:-8], ref[:, 8:-8], atol=2e-5)


def test_istft_length_handling():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(4096).astype(np.float32)
    S = o.stft(x, 512, 128)
    assert o.istft(S, 128).shape == (128 * (S.shape[1] - 1),)
    assert o.istft(S, 128, length=3000).shape == (3000,)
    Su = o.stft(x, 512, 128, center=False)
    assert o.istft(Su, 128, center=False).shape == (512 + 128 * (Su.shape[1] - 1),)
    assert o.istft(Su, 128, center=False, length=5000).shape == (5000,)
    with pytest.raises(ValueError):
        o.istft(np.zeros((2, 2, 257, 4), dtype=np.complex64))


def test_overlap_add_definition():
    """Per-sample form of csrc/metal/overlap_add.metal:27-54 against the loop form."""
    rng = np.random.default_rng(11)
    frames = rng.standard_normal((2, 7, 16))
    win = o.get_window("hann", 16).astype(np.float64)
    hop, L = 4, 16 + 6 * 4
    got = o.overlap_add(frames, win, hop, L)
    want = np.zeros((2, L))
    for i in range(L):
        num = np.zeros(2)
        den = 0.0
        for f in range(max(0, -(-(i - 15) // hop)), min(6, i // hop) + 1):
            num += win[i - f * hop] * frames[:, f, i - f * hop]
            den += win[i - f * hop] ** 2
        want[:, i] = num / max(den, 1e-8)
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_nola():
    assert o.check_nola("hann", 512, 2048)
    assert o.check_nola("hann", 160, 400)
    assert not o.check_nola(np.r_[np.ones(8), np.zeros(8)].astype(np.float32), 16, 16)


def test_magnitude_phase():
    z = np.array([[3 + 4j, -1j], [1 + 0j, -2 + 0j]], dtype=np.complex64)
    np.testing.assert_allclose(o.magnitude(z), [[5, 1], [1, 2]])
    np.testing.assert_allclose(o.phase(z), [[math.atan2(4, 3), -math.pi / 2], [0, math.pi]],
                               rtol=1e-6)


# ----------------------------------------------------------------------------- mel


def test_hz_mel_known_values():
    # MelTests.swift:12-57,194-209
    assert o.hz_to_mel(0.0) == 0.0
    np.testing.assert_allclose(o.hz_to_mel(1000.0), 15.0)
    np.testing.assert_allclose(o.hz_to_mel(200.0 / 3), 1.0)
    np.testing.assert_allclose(o.hz_to_mel(6400.0), 15.0 + 27.0)
    np.testing.assert_allclose(o.hz_to_mel(1000.0, htk=True), 2595.0 * math.log10(1 + 1000 / 700))
    np.testing.assert_allclose(o.hz_to_mel(700.0, htk=True), 2595.0 * math.log10(2.0))
    f = np.array([0.0, 55.0, 440.0, 999.0, 1000.0, 1001.0, 4000.0, 11025.0])
    for htk in (False, True):
        np.testing.assert_allclose(o.mel_to_hz(o.hz_to_mel(f, htk), htk), f, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax,htk", [
    (22050, 2048, 128, 0.0, None, False), (16000, 400, 80, 0.0, 8000.0, False),
    (16000, 400, 128, 0.0, 8000.0, False), (44100, 4096, 128, 0.0, None, False),
    (8000, 256, 40, 0.0, None, False), (48000, 1024, 64, 50.0, 14000.0, True)])
def test_mel_filterbank_matches_transformers(sr, n_fft, n_mels, fmin, fmax, htk):
    audio_utils = pytest.importorskip("transformers.audio_utils")
    ours = o.mel_filterbank(sr, n_fft, n_mels, fmin, fmax, htk, "slaney")
    ref = audio_utils.mel_filter_bank(
        num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin,
        max_frequency=fmax if fmax is not None else sr / 2, sampling_rate=sr, norm="slaney",
        mel_scale="htk" if htk else "slaney").T
    assert ours.dtype == np.float32 and ours.shape == (n_mels, n_fft // 2 + 1)
    # test_stft_parity.py:160-202 compares against librosa's (identical) construction
    np.testing.assert_allclose(ours, ref, rtol=1e-4, atol=1e-6)
    assert (ours >= 0).all()                      # MelTests.swift:78-97
    assert (ours.sum(axis=1) > 0).all()


def test_mel_filterbank_validation_and_norm():
    for kw in (dict(n_mels=0), dict(fmin=-1.0), dict(fmin=9000.0, fmax=8000.0),
               dict(fmax=9000.0), dict(norm="l1")):
        with pytest.raises(ValueError):
            o.mel_filterbank(**{**dict(sr=16000, n_fft=400, n_mels=80), **kw})
    fb = o.mel_filterbank(16000, 400, 40, norm=None)
    assert 0.9 < fb.max() <= 1.0


def test_melspectrogram_composition():
    """mel = basis @ |stft|^p, checked against torch.stft and the float64 path."""
    rng = np.random.default_rng(42)
    x = rng.standard_normal(22050).astype(np.float32)
    win = o.get_window("hann", 2048)
    S = _torch_stft(x, 2048, 512, True, "constant", win)
    basis = o.mel_filterbank(22050, 2048, 128).astype(np.float64)
    for power in (1.0, 2.0):
        ref = basis @ np.abs(S) ** power
        got = o.melspectrogram(x, power=power)
        assert got.dtype == np.float32 and got.shape == (128, 44)
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5 * ref.max())


# ----------------------------------------------------------------------------- dB / DCT / MFCC


def test_db_known_answers():
    # MelTests.swift:153-179
    np.testing.assert_allclose(o.power_to_db(np.array([1.0, 10.0, 100.0], dtype=np.float32)),
                               [0, 10, 20], atol=1e-5)
    np.testing.assert_allclose(o.amplitude_to_db(np.array([1.0, 10.0, 100.0], dtype=np.float32)),
                               [0, 20, 40], atol=1e-5)
    S = np.array([1e-12, 1.0, 1e6], dtype=np.float32)
    np.testing.assert_allclose(o.power_to_db(S), [-20, 0, 60], atol=1e-4)        # top_db clamp
    np.testing.assert_allclose(o.power_to_db(S, top_db=None), [-100, 0, 60], atol=1e-4)
    np.testing.assert_allclose(o.power_to_db(S, ref=np.max, top_db=None), [-160, -60, 0], atol=1e-4)
    with pytest.raises(ValueError):
        o.power_to_db(S, top_db=0.0)
    x = np.array([-30.0, 0.0, 12.5], dtype=np.float32)
    np.testing.assert_allclose(o.power_to_db(o.db_to_power(x), top_db=None), x, atol=1e-4)
    np.testing.assert_allclose(o.amplitude_to_db(o.db_to_amplitude(x), top_db=None), x, atol=1e-4)


@pytest.mark.parametrize("norm", ["ortho", None])
def test_dct_matches_scipy(norm):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((3, 128, 17))
    ref = scipy.fft.dct(x, type=2, axis=1, norm=norm)[:, :40]
    if norm is None:
        ref = ref / 2  # scipy's unnormalised DCT-II carries a factor 2 (mfcc.py:54-57 does not)
    got = o.dct(x, n=40, axis=1, norm=norm)
    np.testing.assert_allclose(got, ref, atol=1e-5)
    assert o.dct(x.astype(np.float32), n=13, axis=1).shape == (3, 13, 17)   # MFCCTests.swift:12-32
    with pytest.raises(ValueError):
        o.dct(x, type=3)


def test_mfcc_composition():
    rng = np.random.default_rng(21)
    x = rng.standard_normal((2, 8000)).astype(np.float32)
    got = o.mfcc(x, sr=16000, n_mfcc=13, n_fft=512, hop_length=128, n_mels=40, lifter=22)
    mel = o.melspectrogram(x, sr=16000, n_fft=512, hop_length=128, n_mels=40, dtype=np.float64)
    db = 10 * np.log10(np.maximum(mel, 1e-10))
    db = np.maximum(db, db.max() - 80.0)
    ref = scipy.fft.dct(db, type=2, axis=1, norm="ortho")[:, :13]
    ref = ref * (1 + 11.0 * np.sin(np.pi * np.arange(1, 14) / 22))[None, :, None]
    assert got.shape == (2, 13, 63) and got.dtype == np.float32
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-3)
    S = np.abs(rng.standard_normal((40, 9))).astype(np.float32)
    np.testing.assert_allclose(o.mfcc(S=S, n_mfcc=5), o.dct(S, n=5, axis=0), atol=0)
    with pytest.raises(ValueError):
        o.mfcc(x, n_mfcc=0)


# ----------------------------------------------------------------------------- model-side callers


@pytest.mark.parametrize("n_mels", [80, 128])
def test_whisper_log_mel_independent(n_mels):
    audio_utils = pytest.importorskip("transformers.audio_utils")
    rng = np.random.default_rng(42)
    x = (0.1 * rng.standard_normal(48000)).astype(np.float32)
    win = o.get_window("hann", 400)
    S = _torch_stft(x, 400, 160, True, "constant", win)
    fb = audio_utils.mel_filter_bank(201, n_mels, 0.0, 8000.0, 16000, norm="slaney",
                                     mel_scale="slaney").T
    ref = np.log10(np.maximum(fb @ np.abs(S) ** 2, 1e-10))
    ref = (np.maximum(ref, ref.max() - 8.0) + 4.0) / 4.0
    got = o.compute_log_mel_spectrogram(x, n_mels=n_mels)
    assert got.shape == (n_mels, 301) and got.dtype == np.float32
    np.testing.assert_allclose(got, ref, atol=1e-4)


def test_whisper_silence_and_global_max():
    out = o.compute_log_mel_spectrogram(np.zeros(16000, dtype=np.float32))
    # log10(float32(1e-10)) is -10 to within one float32 ulp -> (-10 + 4) / 4
    np.testing.assert_allclose(out, -1.5, rtol=0, atol=3e-7)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 16000)).astype(np.float32)
    x[1] *= 1e-4
    both = o.compute_log_mel_spectrogram(x)
    alone = o.compute_log_mel_spectrogram(x[1])
    # a batched call clamps against the max of the whole array (inference.py:112)
    assert both[1].min() > alone.min()
    np.testing.assert_allclose(both[0], o.compute_log_mel_spectrogram(x[0]), atol=0)
    assert o.pad_or_trim(np.zeros(10), 16).shape == (16,)
    assert o.pad_or_trim(np.zeros((2, 20)), 16).shape == (2, 16)


def test_htdemucs_spec_ispec_round_trip():
    rng = np.random.default_rng(42)
    x = (0.1 * rng.standard_normal((1, 2, 44100))).astype(np.float32)
    z = o.htdemucs_spec(x)
    le = math.ceil(44100 / 1024)
    assert z.shape == (1, 2, 2048, le)
    m = np.stack([z.real, z.imag], axis=2).reshape(1, 1, 4, 2048, le)
    y = o.htdemucs_ispec(m, 44100)
    assert y.shape == (1, 1, 2, 44100)
    # the Nyquist bin is dropped on the way, so agreement is ~1e-2 of the 0.1 signal level
    assert np.abs(y[0, 0] - x[0])[:, 4096:-4096].max() < 5e-3


def test_clap_features():
    rng = np.random.default_rng(7)
    x = rng.standard_normal(48000).astype(np.float32)
    mel = o.clap_mel(x)
    assert mel.shape == (101, 64)
    full = o.clap_padded_mel(x)
    assert full.shape == (4, 1001, 64)
    np.testing.assert_array_equal(full[0], full[3])
    np.testing.assert_allclose(o.clap_mel(np.zeros(48000, dtype=np.float32)), -100.0, atol=2e-5)


def test_float32_error_budget():
    """float32 oracle vs its own float64 truth on the headline configuration."""
    rng = np.random.default_rng(1234)
    x = rng.standard_normal(160000).astype(np.float32)
    a = o.compute_log_mel_spectrogram(x, n_mels=128)
    b = o.compute_log_mel_spectrogram(x, n_mels=128, dtype=np.float64)
    assert np.abs(a - b).max() < 1e-5


@pytest.mark.parametrize("shape,out", [((1201, 64), (1001, 64)), ((3000, 64), (1001, 64)), ((7, 5), (3, 5)),
                                        ((1002, 64), (1001, 64)), ((50, 8), (20, 16))])
def test_bilinear_resize_matches_torch(shape, out):
    """The shrink of the CLAP fusion stack (feature_extractor.py:265-272) is torch's bilinear
    interpolate; the restatement reproduces it to float32 rounding of three blends."""
    rng = np.random.default_rng(shape[0])
    img = (rng.standard_normal(shape) * 30 - 40).astype(np.float32)
    want = torch.nn.functional.interpolate(torch.from_numpy(img)[None, None], size=list(out), mode="bilinear",
                                           align_corners=False)[0, 0].numpy()
    got = o.bilinear_resize(img, *out)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)


def test_clap_fusion_structure():
    rng = np.random.default_rng(3)
    audio = rng.standard_normal(480000 + 123456).astype(np.float32)
    full = o.clap_mel(audio)
    stack, starts = o.clap_fusion_mel(audio, np.random.default_rng(11))
    chunk = 480000 // 480 + 1
    assert stack.shape == (4, chunk, 64)
    n = full.shape[0] - chunk + 1
    thirds = np.array_split(np.arange(n), 3)
    for c, s in enumerate(starts):
        assert s in thirds[c]
        assert np.array_equal(stack[1 + c], full[s:s + chunk])
    np.testing.assert_allclose(stack[0], o.bilinear_resize(full, chunk, 64), rtol=0, atol=0)
    # audio that frames to no more than one chunk: four copies of its mel
    short = rng.standard_normal(480000).astype(np.float32)
    stack2, _ = o.clap_fusion_mel(short, np.random.default_rng(0))
    assert np.array_equal(stack2[0], stack2[3]) and stack2.shape == (4, 1001, 64)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_whisper_chain_against_the_transformers_extractor(n_mels):
    """WhisperFeatureExtractor itself (the object applications use) on a 30 s clip: OpenAI's
    front end = reflect padding, last frame dropped; the rest of the chain is the reference's."""
    transformers = pytest.importorskip("transformers")
    rng = np.random.default_rng(5)
    y = (0.1 * rng.standard_normal(480000)).astype(np.float32)
    y[:100000] *= 1e-3
    fe = transformers.WhisperFeatureExtractor(feature_size=n_mels)
    want = fe(y, sampling_rate=16000, return_tensors="np")["input_features"][0]
    mel = o.melspectrogram(y, sr=16000, n_fft=400, hop_length=160, n_mels=n_mels, fmin=0.0, fmax=8000.0,
                           pad_mode="reflect", dtype=np.float64)
    ls = np.log10(np.maximum(mel, 1e-10))[:, :-1]
    got = (np.maximum(ls, ls.max() - 8.0) + 4.0) / 4.0
    assert got.shape == want.shape == (n_mels, 3000)
    assert np.abs(got - want).max() <= 1e-4  # the extractor computes in float32
