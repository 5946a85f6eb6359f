"""GPU parity: stft / istft / stand-alone ops against the CPU oracle.

Tolerances: complex spectra are compared relative to the frame-level scale
(an FFT's float32 error is proportional to the largest bin, not to each bin):
max |S_gpu - S_oracle| <= 2e-6 * max|S_oracle| (the oracle itself sits ~1e-6
from a float64 transform).  The reference's own tolerance for STFT magnitudes
is rtol 1e-4 / atol 1e-5 (tests/parity/test_stft_parity.py:82); both are
asserted.
"""
import numpy as np
import pytest
import torch

import oracle.spectral as o

pytestmark = pytest.mark.gpu

SIZES = [256, 400, 512, 800, 1024, 2048, 4096]


def _sig(n, seed=0, batch=None):
    rng = np.random.default_rng(seed)
    shape = (n,) if batch is None else (batch, n)
    return rng.standard_normal(shape).astype(np.float32)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("n_fft", SIZES)
@pytest.mark.parametrize("pad_mode", ["constant", "reflect", "edge"])
def test_stft_matches_oracle(gpu, n_fft, pad_mode):
    from mlx_audio_amd.primitives import stft

    y = _sig(3 * n_fft + 137, seed=n_fft, batch=3)
    hop = n_fft // 4 if n_fft != 400 else 160
    S = stft(torch.from_numpy(y).to(gpu), n_fft=n_fft, hop_length=hop, pad_mode=pad_mode)
    ref = o.stft(y, n_fft=n_fft, hop_length=hop, pad_mode=pad_mode)
    assert tuple(S.shape) == ref.shape
    assert S.dtype == torch.complex64
    got = S.cpu().numpy()
    assert _rel(got, ref) <= 2e-6
    np.testing.assert_allclose(np.abs(got), np.abs(ref), rtol=1e-4, atol=1e-5 * np.sqrt(n_fft))


def test_stft_1d_odd_length_and_center_false(gpu):
    from mlx_audio_amd.primitives import stft

    y = _sig(22051, seed=1)
    for center in (True, False):
        S = stft(torch.from_numpy(y).to(gpu), n_fft=2048, hop_length=512, center=center)
        ref = o.stft(y, n_fft=2048, hop_length=512, center=center)
        assert tuple(S.shape) == ref.shape
        assert _rel(S.cpu().numpy(), ref) <= 2e-6


def test_stft_odd_batch_stride_unaligned(gpu):
    """Odd signal length makes every second row 8-byte misaligned."""
    from mlx_audio_amd.primitives import stft

    y = _sig(4097, seed=2, batch=4)
    S = stft(torch.from_numpy(y).to(gpu), n_fft=512, hop_length=128)
    assert _rel(S.cpu().numpy(), o.stft(y, n_fft=512, hop_length=128)) <= 2e-6


def test_stft_win_length_and_windows(gpu):
    from mlx_audio_amd.primitives import stft

    y = _sig(8000, seed=3)
    for window, win_length in [("hamming", 300), ("blackman", 400), ("bartlett", 256),
                               ("boxcar", 400)]:
        S = stft(torch.from_numpy(y).to(gpu), n_fft=400, hop_length=160, win_length=win_length,
                 window=window)
        ref = o.stft(y, n_fft=400, hop_length=160, win_length=win_length, window=window)
        assert _rel(S.cpu().numpy(), ref) <= 2e-6
    w = np.random.default_rng(0).random(400).astype(np.float32)
    S = stft(torch.from_numpy(y).to(gpu), n_fft=400, hop_length=100, window=torch.from_numpy(w))
    assert _rel(S.cpu().numpy(), o.stft(y, n_fft=400, hop_length=100, window=w)) <= 2e-6


def test_stft_sine_peak_bin(gpu):
    """440 Hz sine at 22050 Hz, the reference's first parity case."""
    from mlx_audio_amd.primitives import stft

    sr = 22050
    t = np.linspace(0, 1.0, sr, endpoint=False)
    y = np.sin(2 * np.pi * 440.0 * t).astype(np.float32)
    S = stft(torch.from_numpy(y).to(gpu), n_fft=2048, hop_length=512).cpu().numpy()
    ref = o.stft(y, n_fft=2048, hop_length=512)
    np.testing.assert_allclose(np.abs(S), np.abs(ref), rtol=1e-4, atol=1.5e-4)
    assert int(np.abs(S[:, 20]).argmax()) == round(440.0 * 2048 / sr)


def test_stft_errors(gpu):
    from mlx_audio_amd.primitives import stft

    y = torch.zeros(1000, device=gpu)
    with pytest.raises(ValueError, match="hop_length must be positive"):
        stft(y, n_fft=512, hop_length=0)
    with pytest.raises(ValueError, match="must be <= n_fft"):
        stft(y, n_fft=512, win_length=1024)
    with pytest.raises(ValueError, match="Unknown pad_mode"):
        stft(y, n_fft=512, pad_mode="wrap")
    with pytest.raises(ValueError, match="must be >= frame_length"):
        stft(torch.zeros(100, device=gpu), n_fft=512, center=False)
    assert stft(y, n_fft=300).shape == (151, 1 + 1000 // 75)  # no compiled kernel: any-length engine
    # 10 ms at 44.1 kHz is shorter than n_fft but centre padding makes it long enough
    assert stft(torch.zeros(441, device=gpu), n_fft=512).shape == (257, 4)


@pytest.mark.parametrize("n_fft", SIZES)
def test_istft_matches_oracle(gpu, n_fft):
    from mlx_audio_amd.primitives import istft

    rng = np.random.default_rng(n_fft)
    hop = n_fft // 4 if n_fft != 400 else 160
    F = 23
    spec = (rng.standard_normal((2, n_fft // 2 + 1, F)) +
            1j * rng.standard_normal((2, n_fft // 2 + 1, F))).astype(np.complex64)
    ref = o.istft(spec, hop_length=hop)
    got = istft(torch.from_numpy(spec).to(gpu), hop_length=hop).cpu().numpy()
    assert got.shape == ref.shape
    # hann-squared sums are tiny at the very edges, which amplifies round-off there
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    ref_l = o.istft(spec, hop_length=hop, length=hop * 20 + 3)
    got_l = istft(torch.from_numpy(spec).to(gpu), hop_length=hop, length=hop * 20 + 3)
    assert got_l.shape == ref_l.shape
    assert np.abs(got_l.cpu().numpy() - ref_l).max() <= 1e-5 * max(1.0, np.abs(ref_l).max())


@pytest.mark.parametrize("n_fft,hop", [(2048, 512), (400, 160), (4096, 1024), (1024, 480)])
def test_roundtrip_snr(gpu, n_fft, hop):
    """STFT -> iSTFT round trip: SNR >= 100 dB (float32), plus the reference's
    own tolerance rtol 1e-3 / atol 1e-4 (test_stft_parity.py:103-118)."""
    from mlx_audio_amd.primitives import istft, stft

    y = _sig(22050 * 2, seed=42, batch=2)
    yt = torch.from_numpy(y).to(gpu)
    S = stft(yt, n_fft=n_fft, hop_length=hop)
    rec = istft(S, hop_length=hop, length=y.shape[1]).cpu().numpy()
    snr = 10 * np.log10((y.astype(np.float64) ** 2).sum() /
                        ((y.astype(np.float64) - rec) ** 2).sum())
    assert snr >= 100.0, snr
    np.testing.assert_allclose(rec, y, rtol=1e-3, atol=1e-4)


def test_istft_center_false_and_2d(gpu):
    from mlx_audio_amd.primitives import istft, stft

    y = _sig(8192, seed=5)
    S = stft(torch.from_numpy(y).to(gpu), n_fft=512, hop_length=128, center=False)
    ref_S = o.stft(y, n_fft=512, hop_length=128, center=False)
    for length in (None, 5000, 9000):
        ref = o.istft(ref_S, hop_length=128, center=False, length=length)
        got = istft(S, hop_length=128, center=False, length=length).cpu().numpy()
        assert got.shape == ref.shape
        # Without centre padding the squared-window sum falls to ~1e-9 at both
        # ends, so the division amplifies float32 round-off there: tight check
        # on the interior, loose check on the first/last frame.
        n = min(len(ref), 8192)
        inner = slice(512, n - 512)
        assert np.abs(got[inner] - ref[inner]).max() <= 2e-5 * max(1.0, np.abs(ref).max())
        np.testing.assert_allclose(got, ref, rtol=0, atol=5e-3)
    with pytest.raises(ValueError, match="2D or 3D"):
        istft(torch.zeros((2, 2, 257, 4), dtype=torch.complex64, device=gpu))


def test_ext_ops(gpu):
    from mlx_audio_amd.primitives import ext

    y = _sig(1000, seed=7, batch=3)
    yt = torch.from_numpy(y).to(gpu)
    for mode in ("constant", "edge", "reflect"):
        np.testing.assert_array_equal(ext.pad_signal(yt, 200, mode).cpu().numpy(),
                                      o.pad_signal(y, 200, mode))
    assert ext.pad_signal(yt, 0) is not None
    with pytest.raises(ValueError):
        ext.pad_signal(yt, 1000, "reflect")
    np.testing.assert_array_equal(ext.frame_signal(yt, 256, 100).cpu().numpy(),
                                  o.frame_signal(y, 256, 100))
    with pytest.raises(ValueError, match="must be >= frame_length"):
        ext.frame_signal(yt, 2000, 100)
    frames = np.random.default_rng(0).standard_normal((2, 9, 64)).astype(np.float32)
    win = o.get_window("hann", 64)
    got = ext.overlap_add(torch.from_numpy(frames).to(gpu), torch.from_numpy(win).to(gpu), 16,
                          64 + 8 * 16 + 5).cpu().numpy()
    ref = o.overlap_add(frames, win, 16, 64 + 8 * 16 + 5)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(ext.generate_window("hann", 400, True).cpu().numpy(),
                                  o.get_window("hann", 400))
    np.testing.assert_array_equal(ext.get_dct_matrix(13, 40).cpu().numpy(), o.dct_matrix(13, 40))


def test_magnitude_phase(gpu):
    from mlx_audio_amd.primitives import magnitude, phase

    rng = np.random.default_rng(0)
    z = (rng.standard_normal((5, 33)) + 1j * rng.standard_normal((5, 33))).astype(np.complex64)
    zt = torch.from_numpy(z).to(gpu)
    np.testing.assert_allclose(magnitude(zt).cpu().numpy(), np.abs(z), rtol=1e-6)
    np.testing.assert_allclose(phase(zt).cpu().numpy(), np.angle(z), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n_fft,B,F", [
    (4096, 2, 2801),   # HTDemucs framing; more tiles than workgroups: runs start mid-row
    (4096, 3, 2),      # rows shorter than one tile
    (2048, 3, 4099),
    (1024, 5, 12345),
    (1024, 1, 1),
])
def test_fused_istft_equals_two_kernel_path(gpu, n_fft, B, F):
    """For hop = n_fft / 4 mia_istft is one launch that keeps the overlapping frames in LDS:
    it must equal mia_irfft_frames + mia_overlap_add bit for bit, including rows cut
    between workgroups and output lengths that stop short of / run past the last frame."""
    from mlx_audio_amd import _lib
    from mlx_audio_amd._runtime import get_plan, stream_ptr
    from mlx_audio_amd.primitives.stft import padded_window_np

    lib = _lib.lib()
    hop = n_fft // 4
    gen = torch.Generator(device=gpu)
    gen.manual_seed(n_fft + F)
    bins = n_fft // 2 + 1
    spec = torch.view_as_complex(torch.randn((B, F, bins, 2), generator=gen, device=gpu))
    win = padded_window_np("hann", n_fft, n_fft)
    plan = get_plan(spec.device, n_fft, hop, True, "constant", win)
    assert lib.mia_istft_workspace_bytes(plan.handle, B, F) == 0
    win_d = torch.from_numpy(np.array(win, dtype=np.float32)).to(gpu)
    frames = torch.empty((B, F, n_fft), dtype=torch.float32, device=gpu)
    sb, sf, sk = spec.stride()
    full = n_fft + (F - 1) * hop
    for out_len in (full, full - hop // 2 - 3, full + 777):
        with torch.cuda.device(spec.device):
            _lib.check(lib.mia_irfft_frames(plan.handle, spec.data_ptr(), B, F, sb, sk, sf,
                                            frames.data_ptr(), stream_ptr(spec.device)))
            ref = torch.empty((B, out_len), dtype=torch.float32, device=gpu)
            _lib.check(lib.mia_overlap_add(frames.data_ptr(), win_d.data_ptr(), B, F, n_fft, hop,
                                           out_len, 0.5, ref.data_ptr(), stream_ptr(spec.device)))
            got = torch.full((B, out_len), float("nan"), dtype=torch.float32, device=gpu)
            _lib.check(lib.mia_istft(plan.handle, spec.data_ptr(), B, F, sb, sk, sf, out_len, 0.5,
                                     got.data_ptr(), None, 0, stream_ptr(spec.device)))
        assert torch.equal(got, ref), (out_len, float((got - ref).abs().max()))


@pytest.mark.gpu
def test_magnitude_and_phase_of_the_stft_view_skip_the_copy(gpu):
    """magnitude / phase of the strided view `stft` returns (transposing kernel) equal the same
    functions on a contiguous copy (elementwise kernel) bit for bit, for batched and 1-D input
    and tile-unfriendly sizes."""
    import torch

    from mlx_audio_amd.primitives import magnitude, phase, stft

    g = torch.Generator(device=gpu)
    g.manual_seed(0)
    for shape, n_fft, hop in (((3, 9000), 512, 128), ((7001,), 400, 160), ((2, 50000), 2048, 512)):
        y = torch.randn(shape, generator=g, device=gpu)
        C = stft(y, n_fft=n_fft, hop_length=hop)
        assert not C.is_contiguous()
        for fn in (magnitude, phase):
            a = fn(C)
            b = fn(C.contiguous())
            assert a.shape == C.shape and a.is_contiguous()
            assert torch.equal(a, b)


# ---- n_fft without a compiled kernel: the any-length engine (mixed radix / chirp-z) ----------
ANY_N_FFT = [300, 441, 1000, 1536, 8192, 97, 1023, 6]


@pytest.mark.parametrize("n_fft", ANY_N_FFT)
@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
def test_stft_any_length_matches_oracle(gpu, n_fft, pad_mode):
    import oracle.spectral as o
    from mlx_audio_amd.primitives import stft

    rng = np.random.default_rng(n_fft)
    hop = max(1, n_fft // 4 + (1 if n_fft % 3 == 0 else 0))
    y = rng.standard_normal((3, 5 * n_fft + 37)).astype(np.float32)
    got = stft(torch.from_numpy(y).to(gpu), n_fft=n_fft, hop_length=hop, pad_mode=pad_mode).cpu().numpy()
    want = o.stft(y, n_fft=n_fft, hop_length=hop, pad_mode=pad_mode, dtype=np.float64)
    assert got.shape == want.shape == (3, n_fft // 2 + 1, 1 + (y.shape[1] + 2 * (n_fft // 2) - n_fft) // hop)
    assert np.abs(got - want).max() <= 4e-6 * np.abs(want).max()


@pytest.mark.parametrize("n_fft", [300, 441, 1000, 1536, 8192, 97])
def test_istft_any_length_round_trip_and_oracle(gpu, n_fft):
    import oracle.spectral as o
    from mlx_audio_amd.primitives import istft, stft

    rng = np.random.default_rng(1000 + n_fft)
    hop = n_fft // 4
    T = 6 * n_fft + 11
    y = rng.standard_normal((2, T)).astype(np.float32)
    S = stft(torch.from_numpy(y).to(gpu), n_fft=n_fft, hop_length=hop)
    rec = istft(S, hop_length=hop, n_fft=n_fft, length=T).cpu().numpy()
    snr = 10 * np.log10((y.astype(np.float64) ** 2).sum() / ((y - rec).astype(np.float64) ** 2).sum())
    assert snr >= 100.0, snr
    # the inverse alone, on a spectrum that is not the transform of a signal
    Z = (rng.standard_normal((2, n_fft // 2 + 1, 9)) + 1j * rng.standard_normal((2, n_fft // 2 + 1, 9)))
    Z = Z.astype(np.complex64)
    got = istft(torch.from_numpy(Z).to(gpu), hop_length=hop, n_fft=n_fft).cpu().numpy()
    want = o.istft(Z, hop_length=hop, n_fft=n_fft, dtype=np.float64)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 4e-6 * np.abs(want).max()


@pytest.mark.parametrize("n_fft,n_mels", [(300, 20), (441, 40), (1000, 64), (1536, 80)])
def test_mel_and_mfcc_any_length(gpu, n_fft, n_mels):
    import oracle.spectral as o
    from mlx_audio_amd.primitives import melspectrogram, mfcc

    rng = np.random.default_rng(n_fft + n_mels)
    y = rng.standard_normal((2, 16000)).astype(np.float32)
    yd = torch.from_numpy(y).to(gpu)
    got = melspectrogram(yd, sr=16000, n_fft=n_fft, hop_length=n_fft // 3, n_mels=n_mels).cpu().numpy()
    want = np.stack([o.melspectrogram(y[b], sr=16000, n_fft=n_fft, hop_length=n_fft // 3, n_mels=n_mels)
                     for b in range(2)])
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5 * want.max())
    gm = mfcc(yd, sr=16000, n_mfcc=13, n_fft=n_fft, hop_length=n_fft // 3, n_mels=n_mels,
              per_clip=True).cpu().numpy()
    wm = np.stack([o.mfcc(y[b], sr=16000, n_mfcc=13, n_fft=n_fft, hop_length=n_fft // 3, n_mels=n_mels)
                   for b in range(2)])
    np.testing.assert_allclose(gm, wm, rtol=1e-5, atol=2e-3)


def test_any_length_plans_reject_the_entry_points_without_workspace(gpu):
    from mlx_audio_amd import _lib
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives.windows import padded_window_np

    lib = _lib.lib()
    assert lib.mia_fft_size_supported(422) == 0 and lib.mia_fft_size_supported(400) == 1
    # a size with a large prime factor (422 = 2 x 211) runs on the whole-row engine, which needs its workspace
    plan = get_plan(gpu, 422, 110, True, "constant", padded_window_np("hann", 422, 422))
    y = torch.zeros((1, 2000), device=gpu)
    out = torch.empty((1, plan.num_frames(2000), 212), dtype=torch.complex64, device=gpu)
    assert lib.mia_stft_workspace_bytes(plan.handle, 1, 2000) > 0
    rc = lib.mia_stft(plan.handle, y.data_ptr(), 1, 2000, 1.0, out.data_ptr(), None)
    assert rc == _lib.MIA_ERR_INVALID and b"mia_stft_ws" in lib.mia_last_error()
    # a 7-smooth n_fft (even up to 2048, odd up to 1023) has no compiled plan either but is served by
    # the per-wave kernel: no workspace, the plain entry point works
    plan = get_plan(gpu, 300, 75, True, "constant", padded_window_np("hann", 300, 300))
    assert lib.mia_fft_size_supported(300) == 0
    assert lib.mia_stft_workspace_bytes(plan.handle, 1, 1000) == 0
    y = torch.zeros((1, 1000), device=gpu)
    out = torch.empty((1, plan.num_frames(1000), 151), dtype=torch.complex64, device=gpu)
    _lib.check(lib.mia_stft(plan.handle, y.data_ptr(), 1, 1000, 1.0, out.data_ptr(), None))
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0


WAVE_SIZES = [4, 6, 24, 98, 300, 320, 480, 640, 882, 960, 1000, 1200, 1280, 1536, 1600, 1764, 1920, 2000, 2016, 2048 - 2 * 7,
              3, 9, 105, 441, 675, 945,   # ... odd sizes, transformed unpacked
              8, 16, 32, 64, 128, 12, 18, 50]  # ... and short ones: several frames side by side in a wave


@pytest.mark.parametrize("n_fft", WAVE_SIZES)
def test_per_wave_transform_sizes(gpu, n_fft):
    """n_fft with 7-smooth (half) lengths, even up to 2048 and odd up to 1023 (radices 2 ... 16 in
    every mix: 882 = 2 (7 7 3 3), 960 = 2 (12 10 4), 2000 = 2 (10 10 10), 441 = 7 7 3 3, 945 = 7 5 3 3 3): spectrum against float64, every
    padding mode, centre off, a short window, ragged frame counts; and against the whole-row
    engine the size ran on before (plan option wave_fft = 0)."""
    from mlx_audio_amd import _lib
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives import stft
    from mlx_audio_amd.primitives.windows import padded_window_np

    hop = max(1, n_fft // 4)
    g = torch.Generator(device=gpu).manual_seed(n_fft)
    y = torch.randn((3, 11 * n_fft + 5), device=gpu, generator=g)
    y[1, : 3 * n_fft] *= 1e-3
    win = torch.hann_window(n_fft, periodic=True, dtype=torch.float64, device=gpu)
    for pad_mode in ("reflect", "constant"):
        got = stft(y, n_fft=n_fft, hop_length=hop, pad_mode=pad_mode)
        ref = torch.stft(y.double(), n_fft, hop_length=hop, window=win, center=True, pad_mode=pad_mode,
                         return_complex=True)
        assert got.shape == ref.shape
        assert float((got - ref).abs().max() / ref.abs().max()) <= 2e-6, pad_mode
    if n_fft >= 24:
        got = stft(y, n_fft=n_fft, hop_length=hop + 1, center=False, win_length=n_fft - 3)
        w2 = torch.zeros(n_fft, dtype=torch.float64, device=gpu)
        lo = (n_fft - (n_fft - 3)) // 2
        w2[lo: lo + n_fft - 3] = torch.hann_window(n_fft - 3, periodic=True, dtype=torch.float64, device=gpu)
        ref = torch.stft(y.double(), n_fft, hop_length=hop + 1, window=w2, center=False, return_complex=True)
        assert float((got - ref).abs().max() / ref.abs().max()) <= 2e-6
    # the same plan on the whole-row engine, forward and inverse
    from mlx_audio_amd.primitives import istft

    lib = _lib.lib()
    plan = get_plan(y.device, n_fft, hop, True, "edge", padded_window_np("hann", n_fft, n_fft))
    a = stft(y, n_fft=n_fft, hop_length=hop, pad_mode="edge")
    xa = istft(a, hop_length=hop, n_fft=n_fft, length=y.shape[1]) if n_fft >= 24 else None
    try:
        _lib.check(lib.mia_plan_set_option(plan.handle, b"wave_fft", 0))
        assert lib.mia_stft_workspace_bytes(plan.handle, 3, y.shape[1]) > 0
        b = stft(y, n_fft=n_fft, hop_length=hop, pad_mode="edge")
        xb = istft(a, hop_length=hop, n_fft=n_fft, length=y.shape[1]) if n_fft >= 24 else None
    finally:
        lib.mia_plan_set_option(plan.handle, b"wave_fft", 1)
    assert float((a - b).abs().max() / b.abs().max()) <= 3e-6
    if xa is not None:  # hann at hop n_fft / 4 (or next to it) inverts; the edges lack overlap
        assert float((xa - xb).abs().max()) <= 2e-5
        m = n_fft
        assert float((xa[:, m:-m] - y[:, m:-m]).abs().max()) <= 3e-5
        one = istft(a[2], hop_length=hop, n_fft=n_fft, length=y.shape[1])
        assert torch.equal(one, xa[2])


@pytest.mark.parametrize("n_fft,hop", [(1024, 256), (2048, 512), (4096, 1024), (512, 128), (400, 160)])
def test_istft_of_a_frames_contiguous_spectrum(gpu, n_fft, hop):
    """A contiguous (batch, bins, frames) array - the reference's layout - gives bit for bit what
    the (batch, frames, bins) view stft() returns gives (the fused sizes transpose it first:
    mia_transpose_spectrum), odd frame counts and a single row included."""
    from mlx_audio_amd import _lib
    from mlx_audio_amd.primitives import istft, stft

    g = torch.Generator(device=gpu)
    g.manual_seed(n_fft)
    for rows, T in ((3, 9 * n_fft + 37), (1, 2 * n_fft)):
        y = torch.randn((rows, T), generator=g, device=gpu)
        S = stft(y, n_fft=n_fft, hop_length=hop)
        Sc = S.contiguous()
        assert Sc.stride(2) == 1
        a = istft(S, hop_length=hop, length=T)
        b = istft(Sc, hop_length=hop, length=T)
        assert torch.equal(a, b)
        assert torch.equal(istft(Sc[0], hop_length=hop, length=T), a[0])
    lib = _lib.lib()
    s = torch.cuda.current_stream(gpu).cuda_stream
    out = torch.empty((rows, S.shape[2], S.shape[1]), dtype=torch.complex64, device=gpu)
    _lib.check(lib.mia_transpose_spectrum(Sc.data_ptr(), rows, S.shape[1], S.shape[2], out.data_ptr(), s))
    assert torch.equal(torch.view_as_real(out), torch.view_as_real(S.transpose(1, 2).contiguous()))
    assert lib.mia_transpose_spectrum(Sc.data_ptr(), rows, S.shape[1], S.shape[2], Sc.data_ptr(), s) == _lib.MIA_ERR_INVALID



@pytest.mark.parametrize("sr,n_fft,hop,win", [(48000, 1920, 480, 960), (16000, 640, 160, 320)])
def test_enhance_model_framings_round_trip(gpu, sr, n_fft, hop, win):
    """The framings of the reference's speech-enhancement model (models/enhance/config.py:28-31,
    63-68; its _stft / _istft call stft and istft with a window half the FFT size, model.py:262-
    291): they run on the per-wave transform's compiled-in plans.  Spectrum against the float64
    oracle, and the inverse gives the signal back."""
    from mlx_audio_amd.primitives import istft, stft

    rng = np.random.default_rng(n_fft)
    y = rng.standard_normal((3, sr // 2 + 17)).astype(np.float32)
    yd = torch.from_numpy(y).to(gpu)
    kw = dict(n_fft=n_fft, hop_length=hop, win_length=win, window="hann", center=True)
    C = stft(yd, **kw)
    want = o.stft(y.astype(np.float64), **kw)
    scale = np.abs(want).max()
    assert np.abs(C.cpu().numpy() - want).max() <= 3e-6 * scale
    back = istft(C, hop_length=hop, win_length=win, n_fft=n_fft, window="hann", center=True, length=y.shape[1])
    # the half-size window covers every sample twice (hop = win / 2): the round trip is exact to round-off
    assert (back - yd).abs().max().item() <= 5e-6 * float(np.abs(y).max())
    assert np.abs(back.cpu().numpy() - o.istft(want, hop_length=hop, win_length=win, n_fft=n_fft, length=y.shape[1])).max() <= 1e-5


@pytest.mark.parametrize("n_fft,hop,win", [(1920, 480, 960), (640, 160, 320), (480, 120, 480), (960, 240, 960),
                                           (882, 441, 882), (1200, 300, 1200), (1536, 192, 1536), (1764, 500, 1764),
                                           (960, 480, 960), (480, 77, 300),
                                           # the small sizes with a compiled forward plan: inverse-only per-wave plans
                                           (512, 128, 512), (400, 100, 400), (256, 64, 256), (800, 200, 800),
                                           (400, 160, 400), (512, 200, 400), (1024, 384, 1024), (2048, 768, 2048),
                                           (1024, 128, 600)])
def test_fused_per_wave_inverse_equals_the_two_kernel_form(gpu, n_fft, hop, win):
    """istft on the per-wave transform's compiled-in plans keeps the overlap in an LDS ring; the
    same plan with option wave_istft_fused = 0 writes the frame workspace and runs the
    overlap-add kernel.  Same overlap-add (additions in the same order, same envelope); the
    transforms are two instantiations of the same passes (in place / between two buffers) whose
    multiply-adds the compiler contracts differently, so the samples agree to float32 round-off
    of the transform (3e-6 of the signal's peak), for rows long enough that a row is cut into
    several runs, for one to three frames, for a requested length beyond the frames' reach and
    for hops that do not divide the frame; and against the oracle."""
    from mlx_audio_amd import _lib
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives import istft, stft
    from mlx_audio_amd.primitives.windows import padded_window_np

    lib = _lib.lib()
    plan = get_plan(torch.device(gpu), n_fft, hop, True, "constant", padded_window_np("hann", win, n_fft))
    rng = np.random.default_rng(n_fft + hop)
    for rows, T in ((3, 100 * hop + 17), (2, 400 * hop + 5), (1, n_fft), (2, n_fft + hop), (300, 8 * hop + 1)):
        y = rng.standard_normal((rows, T)).astype(np.float32)
        C = stft(torch.from_numpy(y).to(gpu), n_fft=n_fft, hop_length=hop, win_length=win)
        kw = dict(hop_length=hop, win_length=win, n_fft=n_fft)
        assert lib.mia_istft_workspace_bytes(plan.handle, rows, C.shape[-1]) == 0
        fused = [istft(C, length=L, **kw) for L in (T, None, T + 3 * n_fft)]
        fused.append(istft(C, center=False, **kw))
        _lib.check(lib.mia_plan_set_option(plan.handle, b"wave_istft_fused", 0))
        try:
            assert lib.mia_istft_workspace_bytes(plan.handle, rows, C.shape[-1]) > 0
            two = [istft(C, length=L, **kw) for L in (T, None, T + 3 * n_fft)]
            two.append(istft(C, center=False, **kw))
        finally:
            lib.mia_plan_set_option(plan.handle, b"wave_istft_fused", 1)
        # Under a single window foot (the ends of an uncentred result, the tail of a length
        # beyond the signal) the division by an envelope of 1e-4 .. 1e-8 magnifies the transforms'
        # last bits: tight in the body, loose there.
        tol = 3e-6 * float(np.abs(y).max())
        for a, b in zip(fused, two):
            assert a.shape == b.shape
            if T > 3 * n_fft:
                assert (a - b)[:, n_fft:T - n_fft].abs().max().item() <= tol
            assert torch.allclose(a, b, rtol=2e-2, atol=1e-2)
        assert (fused[0] - two[0]).abs().max().item() <= tol   # centred, cut to the signal: no foot at all
        if rows <= 3 and hop * 2 <= win:   # the window overlaps itself: the signal comes back
            want = o.istft(o.stft(y.astype(np.float64), n_fft=n_fft, hop_length=hop, win_length=win),
                           hop_length=hop, win_length=win, n_fft=n_fft, length=T)
            assert np.abs(fused[0].cpu().numpy() - want).max() <= 2e-5


def test_fused_per_wave_inverse_random_framings(gpu):
    """Thirty random framings on the compiled-in per-wave plans (hop anywhere from n_fft / 8 to
    n_fft, window anywhere up to n_fft, row counts and lengths at random, centred or not, a
    strided spectrum): fused inverse against the float64 oracle wherever the window envelope is
    not tiny, and against the two-kernel form's shape."""
    from mlx_audio_amd.primitives import istft, stft

    rng = np.random.default_rng(2026)
    sizes = [320, 480, 640, 882, 960, 1200, 1536, 1764, 1920]
    for case in range(30):
        n_fft = int(rng.choice(sizes))
        hop = int(rng.integers(max(1, n_fft // 8), n_fft // 2 + 1))
        win = int(rng.integers(max(2 * hop, n_fft // 4), n_fft + 1))
        rows = int(rng.integers(1, 6))
        T = int(rng.integers(n_fft, 40 * n_fft))
        center = bool(rng.integers(0, 2))
        y = rng.standard_normal((rows, T)).astype(np.float32)
        kw = dict(n_fft=n_fft, hop_length=hop, win_length=win, center=center)
        C = stft(torch.from_numpy(y).to(gpu), **kw)
        if case % 3 == 0:   # frames contiguous, as the reference lays its spectrum out
            C = C.contiguous()
        want = o.istft(o.stft(y.astype(np.float64), **kw), hop_length=hop, win_length=win, n_fft=n_fft,
                       center=center, length=T if center else None)
        got = istft(C, hop_length=hop, win_length=win, n_fft=n_fft, center=center,
                    length=T if center else None).cpu().numpy()
        assert got.shape == want.shape, (case, n_fft, hop, win, rows, T, center)
        # where at least one window of the sum is not near its foot the two agree to round-off
        w = o.get_window("hann", win)
        env = np.zeros(n_fft + hop * (C.shape[-1] - 1))
        pw = np.zeros(n_fft)
        pw[(n_fft - win) // 2:(n_fft - win) // 2 + win] = w ** 2
        for f in range(C.shape[-1]):
            env[f * hop:f * hop + n_fft] += pw
        env = env[n_fft // 2:n_fft // 2 + want.shape[1]] if center else env[:want.shape[1]]
        ok = env > 1e-2
        assert ok.mean() > 0.5
        assert np.abs(got - want)[:, ok].max() <= 2e-4, (case, n_fft, hop, win, rows, T, center)


def test_compiled_in_plans_are_the_ones_taken(gpu):
    """The launcher takes a compiled-in plan only when the planner's choice for the n_fft is
    exactly that plan: a change of the planner's cost model must not silently send these sizes
    back to the run-time form.  And the framings documented as workspace-free are."""
    from mlx_audio_amd import _lib
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives.windows import padded_window_np

    lib = _lib.lib()
    dev = torch.device(gpu)

    def plan(n_fft, hop):
        return get_plan(dev, n_fft, hop, True, "constant", padded_window_np("hann", n_fft, n_fft))

    for n_fft in (320, 441, 480, 640, 882, 960, 1000, 1200, 1280, 1536, 1600, 1764, 1920, 2000):
        assert lib.mia_plan_get_info(plan(n_fft, n_fft // 4).handle, b"wave_fixed_plan") == 1, n_fft
    for n_fft in (300, 2016, 98):
        assert lib.mia_plan_get_info(plan(n_fft, n_fft // 2).handle, b"wave_fixed_plan") == 0, n_fft
    assert lib.mia_plan_get_info(plan(1024, 256).handle, b"wave_fixed_plan") == -1
    fused = [(1024, 256), (2048, 512), (4096, 1024), (1920, 480), (640, 160), (480, 120), (960, 300),
             (256, 64), (400, 160), (512, 128), (800, 200), (1024, 384), (2048, 768), (1764, 441)]
    for n_fft, hop in fused:
        assert lib.mia_plan_get_info(plan(n_fft, hop).handle, b"istft_fused") == 1, (n_fft, hop)
    for n_fft, hop in ((320, 80), (441, 110), (4096, 512), (300, 75), (1920, 100)):
        assert lib.mia_plan_get_info(plan(n_fft, hop).handle, b"istft_fused") == 0, (n_fft, hop)
