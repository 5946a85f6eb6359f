"""Sample-rate conversion on the GPU against the oracle (SciPy / NumPy, the reference's
own arithmetic for this path): bit-exact, through the C ABI."""

import numpy as np
import pytest
import torch

import oracle.resample as orc
from mlx_audio_amd import _lib
from mlx_audio_amd.primitives import resample, resample_poly

pytestmark = pytest.mark.gpu

RATIOS = [(1, 2), (2, 1), (3, 1), (1, 3), (147, 160), (160, 147), (160, 441), (441, 160),
          (5, 7), (2, 3), (1, 6), (1, 160), (320, 441)]


def _dev(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


@pytest.mark.parametrize("up,down", RATIOS)
def test_resample_poly_is_bit_identical_to_scipy(gpu, up, down):
    rng = np.random.default_rng(100 * up + down)
    for n in (1, 2, 3, 63, 64, 1000, 4097, 30011):
        x = rng.standard_normal((3, n)).astype(np.float32)
        want = orc.resample_poly(x, up, down)
        got = resample_poly(_dev(x, gpu), up, down).cpu().numpy()
        assert got.shape == want.shape and got.dtype == np.float32
        np.testing.assert_array_equal(got, want, err_msg=f"{up}/{down} n={n}")


def test_resample_poly_reduces_the_ratio_and_takes_1d_and_axis(gpu):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(5000).astype(np.float32)
    np.testing.assert_array_equal(resample_poly(_dev(x, gpu), 4, 8).cpu().numpy(),
                                  orc.resample_poly(x, 1, 2))
    x2 = rng.standard_normal((700, 3)).astype(np.float32)
    got = resample_poly(_dev(x2, gpu), 3, 2, axis=0)
    assert got.shape == (1050, 3)
    np.testing.assert_array_equal(got.cpu().numpy(), orc.resample_poly(x2, 3, 2, axis=0))
    # host arrays and other dtypes are taken too, as the reference takes anything array-like
    np.testing.assert_array_equal(resample_poly(x.astype(np.float64), 1, 2).cpu().numpy(),
                                  orc.resample_poly(x, 1, 2))


@pytest.mark.parametrize("padtype", ["edge", "reflect", "symmetric", "wrap", "line", "smooth",
                                     "antisymmetric", "antireflect"])
@pytest.mark.parametrize("up,down", [(1, 2), (3, 1), (160, 441), (147, 160)])
def test_resample_poly_extension_modes(gpu, padtype, up, down):
    rng = np.random.default_rng(11)
    for n in (1, 2, 7, 50, 3001):   # rows far shorter than the filter wrap around many times
        x = (rng.standard_normal((2, n)) + 3.0).astype(np.float32)
        # SciPy divides by the reflection period 2*(n-1) = 0 for a one-sample row and the
        # process dies with SIGFPE (reflect, antireflect), or reads past the row (smooth) or
        # divides 0 by 0 (line); here that row is extended by its only sample.
        one = n == 1 and padtype in ("reflect", "antireflect", "smooth", "line")
        ref_mode = "edge" if one else padtype
        want = orc.resample_poly(x, up, down, padtype=ref_mode)
        got = resample_poly(_dev(x, gpu), up, down, padtype=padtype).cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg=f"{padtype} {up}/{down} n={n}")


@pytest.mark.parametrize("padtype", ["mean", "median", "minimum", "maximum"])
def test_resample_poly_background_modes(gpu, padtype):
    rng = np.random.default_rng(12)
    for n in (9, 10, 2001):
        x = (rng.standard_normal((2, n)) + 5.0).astype(np.float32)
        want = orc.resample_poly(x, 2, 3, padtype=padtype)
        got = resample_poly(_dev(x, gpu), 2, 3, padtype=padtype).cpu().numpy()
        # the row statistic is a device reduction (summation order differs for the mean)
        tol = 0.0 if padtype in ("minimum", "maximum") else 2e-6
        np.testing.assert_allclose(got, want, rtol=0, atol=tol)
    with pytest.raises(ValueError, match="padtype must be one of"):
        resample_poly(_dev(x, gpu), 2, 3, padtype="bogus")


def test_resample_poly_nonfinite_samples_stay_local(gpu):
    x = np.random.default_rng(2).standard_normal((2, 4000)).astype(np.float32)
    x[0, 1000] = np.nan
    x[1, 3000] = np.inf
    want = orc.resample_poly(x, 160, 441)
    got = resample_poly(_dev(x, gpu), 160, 441).cpu().numpy()
    np.testing.assert_array_equal(got, want)   # NaN positions and payload-free equality
    assert np.isfinite(got[0, :300]).all()


SHARED_RATIOS = [(160, 441), (1, 3), (3, 1), (1, 2), (2, 1), (160, 147), (2, 3),
                 (1, 6), (5, 7),
                 # lane spacings that are a multiple of four samples: the row-per-lane layout
                 # (long filters only: 48 -> 44.1 kHz stays on the per-output kernel)
                 (1, 4), (3, 8),
                 # integer upsampling by a large factor: the table of step constants is most
                 # of the workgroup's LDS
                 (32, 1), (48, 1), (147, 1), (100, 3)]


def _taps_per_phase(n_in, up, down):
    from mlx_audio_amd.primitives.resample import _kaiser_lowpass, _poly_layout
    h, half_len = _kaiser_lowpass(up, down)
    pre, post, skip, n_out = _poly_layout(n_in, up, down, half_len, len(h))
    return -(-(pre + len(h) + post) // up), n_out


@pytest.mark.parametrize("up,down", SHARED_RATIOS)
def test_shared_window_kernel_is_bit_identical_to_scipy(gpu, up, down):
    """Rows long enough for the kernel that shares a lane's sample window between adjacent
    outputs (several runs per row, a ragged last run, rows at every 4-byte phase of a 16-byte
    line), against SciPy bit for bit, with zero and with signal extension at the ends."""
    rng = np.random.default_rng(7 * up + down)
    n = 100003 if down >= up else 60001
    P, n_out = _taps_per_phase(n, up, down)
    assert _lib.lib().mia_upfirdn_margin(up, down, P, n_out) > 0   # the path under test
    x = rng.standard_normal((5, n)).astype(np.float32)
    for padtype in ("constant", "reflect", "wrap", "line", "antireflect"):
        want = orc.resample_poly(x, up, down, padtype=padtype)
        got = resample_poly(_dev(x, gpu), up, down, padtype=padtype).cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg=f"{up}/{down} {padtype}")


def test_shared_window_kernel_with_nonfinite_samples(gpu):
    """A product of an infinite sample with a zero tap is NaN, not zero: runs that hold a
    non-finite sample must give exactly what the per-output sum gives (the other runs too)."""
    rng = np.random.default_rng(21)
    x = rng.standard_normal((3, 120000)).astype(np.float32)
    x[0, 50000] = np.inf
    x[1, 119990] = np.nan
    x[1, 3] = -np.inf
    for up, down in ((160, 441), (1, 3), (3, 1), (1, 4)):
        want = orc.resample_poly(x, up, down)
        got = resample_poly(_dev(x, gpu), up, down).cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg=f"{up}/{down}")
        assert np.isfinite(got[2]).all() and np.isfinite(got[0, :1000]).all()


def test_shared_window_and_per_output_kernels_agree_through_the_c_abi(gpu):
    from mlx_audio_amd.primitives.resample import (_kaiser_lowpass, _phase_table_dev,
                                                     _poly_layout)
    lib = _lib.lib()
    up, down, rows = 160, 441, 4
    g = torch.Generator(device=gpu)
    g.manual_seed(3)
    for n_in in (44100 * 3, 44100 * 3 + 1):   # odd row length: rows start at every 4-byte phase
        x = torch.randn((rows, n_in), generator=g, device=gpu, dtype=torch.float32)
        h, half_len = _kaiser_lowpass(up, down)
        pre, post, skip, n_out = _poly_layout(n_in, up, down, half_len, len(h))
        P = -(-(pre + len(h) + post) // up)
        margin = lib.mia_upfirdn_margin(up, down, P, n_out)
        assert margin > 0
        plain = _phase_table_dev(up, down, pre, post, x.device)
        padded = _phase_table_dev(up, down, pre, post, x.device, margin)
        assert padded.shape == (up, P + 2 * margin)
        a = torch.empty((rows, n_out), device=gpu)
        b = torch.full((rows, n_out), 7.0, device=gpu)
        stream = torch.cuda.current_stream(gpu).cuda_stream
        _lib.check(lib.mia_upfirdn(x.data_ptr(), rows, n_in, plain.data_ptr(), up, down, P, skip,
                                   n_out, 0, a.data_ptr(), stream))
        _lib.check(lib.mia_upfirdn_shared(x.data_ptr(), rows, n_in, padded.data_ptr(), margin, up,
                                          down, P, skip, n_out, 0, b.data_ptr(), stream))
        assert torch.equal(a, b)
    # a table with another margin, and a ratio the kernel does not take, are refused
    assert lib.mia_upfirdn_shared(x.data_ptr(), rows, n_in, padded.data_ptr(), margin + 1, up, down,
                                  P, skip, n_out, 0, b.data_ptr(), stream) == _lib.MIA_ERR_INVALID
    assert lib.mia_upfirdn_margin(1, 8, 161, 10 ** 6) == 0
    assert lib.mia_upfirdn_shared(x.data_ptr(), rows, n_in, padded.data_ptr(), 20, 1, 8, 161,
                                  skip, n_out, 0, b.data_ptr(), stream) == _lib.MIA_ERR_UNSUPPORTED
    assert lib.mia_upfirdn_margin(up, down, P, 1000) == 0    # short rows: the per-output kernel


@pytest.mark.parametrize("orig,target", [(44100, 16000), (16000, 44100), (48000, 16000),
                                         (22050, 16000), (8000, 48000), (16000, 15999)])
def test_linear_is_bit_identical_to_numpy(gpu, orig, target):
    rng = np.random.default_rng(orig + target)
    for n in (1, 2, 3, 101, 44100, 100003):
        x = rng.standard_normal((2, n)).astype(np.float32)
        for fix in (True, False):
            for scale in (False, True):
                want = orc.resample(x, orig, target, res_type="linear", fix=fix, scale=scale)
                got = resample(_dev(x, gpu), orig, target, res_type="linear", fix=fix, scale=scale)
                got = got.cpu().numpy()
                assert got.shape == want.shape
                np.testing.assert_array_equal(got, want, err_msg=f"n={n} fix={fix} scale={scale}")


def test_linear_1d_axis_and_same_length(gpu):
    rng = np.random.default_rng(8)
    x = rng.standard_normal(1234).astype(np.float32)
    np.testing.assert_array_equal(resample(_dev(x, gpu), 3, 2, res_type="linear").cpu().numpy(),
                                  orc.resample(x, 3, 2, res_type="linear"))
    x2 = rng.standard_normal((500, 2)).astype(np.float32)
    got = resample(_dev(x2, gpu), 1, 2, res_type="linear", axis=0)
    np.testing.assert_array_equal(got.cpu().numpy(), orc.resample(x2, 1, 2, res_type="linear", axis=0))
    # a ratio that rounds to the same length returns the samples unchanged (resample.py:174-175)
    same = resample(_dev(x, gpu), 100000, 100001, res_type="linear")
    np.testing.assert_array_equal(same.cpu().numpy(), x)


def test_full_size_clip_44k1_to_16k(gpu):
    """30 s at 44.1 kHz -> 16 kHz (up 160, down 441), the conversion in front of the Whisper
    front-end, on a batch; three rows are checked against SciPy, all rows against each other
    by linearity: resample(a + 2 b) == resample(a) + 2 resample(b) up to float32 rounding."""
    n = 44100 * 30
    g = torch.Generator(device=gpu)
    g.manual_seed(7)
    x = torch.randn((16, n), generator=g, device=gpu, dtype=torch.float32)
    y = resample_poly(x, 16000, 44100)
    assert y.shape == (16, 480000)
    xc = x[[0, 7, 15]].cpu().numpy()
    np.testing.assert_array_equal(y[[0, 7, 15]].cpu().numpy(), orc.resample_poly(xc, 160, 441))
    mix = resample_poly(x[:8] + 2.0 * x[8:], 160, 441)
    err = (mix - (y[:8] + 2.0 * y[8:])).abs().max().item()
    assert err < 2e-5, err


FFT_TOL = 2e-6  # of the row's peak: float32 FFT round-off (SciPy's own float32 path: <= 7e-7)

FFT_SIZES = [(1, 3), (2, 3), (3, 2), (2, 1), (7, 5), (16, 8), (8, 16), (17, 16), (441, 160),
             (160, 441), (1000, 999), (1001, 1000), (4410, 1600), (44100, 16000),
             (100003, 36283), (65536, 32768), (32768, 65536),
             # rows long enough for the two-super-pass plans (halves 220 500, 240 000, 661 500)
             (441000, 160000), (480000, 441000), (1323000, 480000)]


@pytest.mark.parametrize("n,m", FFT_SIZES)
def test_fourier_method_matches_scipy(gpu, n, m):
    """Any pair of lengths (primes included); checked against SciPy in float64 and float32."""
    import scipy.signal as ss

    rng = np.random.default_rng(n * 7 + m)
    x = rng.standard_normal((3, n)).astype(np.float32)
    got = resample(_dev(x, gpu), n, m, res_type="fft").cpu().numpy()
    want = orc.resample(x, n, m, res_type="fft")
    truth = ss.resample(x.astype(np.float64), m, axis=-1)
    assert got.shape == want.shape and got.dtype == np.float32
    peak = np.abs(truth).max()
    assert np.abs(got - truth).max() <= FFT_TOL * peak
    assert np.abs(got - want).max() <= FFT_TOL * peak


def test_fourier_method_options_and_shapes(gpu):
    rng = np.random.default_rng(77)
    x = rng.standard_normal((2, 4409)).astype(np.float32)
    for fix in (True, False):
        for scale in (False, True):
            want = orc.resample(x, 44100, 16000, fix=fix, scale=scale)
            got = resample(_dev(x, gpu), 44100, 16000, fix=fix, scale=scale).cpu().numpy()  # default = fft
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= FFT_TOL * np.abs(want).max()
    one = resample(_dev(x[0], gpu), 44100, 48000)
    assert one.shape == orc.resample(x[0], 44100, 48000).shape
    assert np.abs(one.cpu().numpy() - orc.resample(x[0], 44100, 48000)).max() <= FFT_TOL * 5
    xt = np.ascontiguousarray(x.T)
    got = resample(_dev(xt, gpu), 2, 1, axis=0)
    want = orc.resample(xt, 2, 1, axis=0)
    assert got.shape == want.shape == (2204, 2)   # numpy.round(2204.5) = 2204
    assert np.abs(got.cpu().numpy() - want).max() <= FFT_TOL * 5
    with pytest.raises(ValueError, match="no output samples"):
        resample(_dev(x[:, :1], gpu), 48000, 8000)


def test_fourier_method_known_answers(gpu):
    # a constant stays that constant; a sine on a bin of both grids keeps amplitude and frequency
    c = np.full((1, 3000), 0.75, np.float32)
    got = resample(_dev(c, gpu), 3000, 1234).cpu().numpy()
    np.testing.assert_allclose(got, 0.75, rtol=0, atol=2e-6)
    n, m, cycles = 44100, 16000, 440
    t = np.arange(n) / n
    x = np.sin(2 * np.pi * cycles * t).astype(np.float32)[None]
    got = resample(_dev(x, gpu), n, m).cpu().numpy()[0]
    want = np.sin(2 * np.pi * cycles * np.arange(m) / m)
    np.testing.assert_allclose(got, want, rtol=0, atol=5e-6)
    # content above the new Nyquist frequency is removed, not aliased
    hi = np.sin(2 * np.pi * 12000 * t).astype(np.float32)[None]
    assert np.abs(resample(_dev(hi, gpu), n, m).cpu().numpy()).max() < 1e-5


def test_fourier_method_row_chunks_are_independent(gpu, monkeypatch):
    import importlib

    mod = importlib.import_module("mlx_audio_amd.primitives.resample")
    rng = np.random.default_rng(9)
    x = _dev(rng.standard_normal((5, 30000)).astype(np.float32), gpu)
    whole = resample(x, 30000, 10007)
    per_row = _lib.lib().mia_resample_fft_workspace_bytes(1, 30000, 10007)
    monkeypatch.setattr(mod, "_FFT_WORKSPACE_BYTES", 2 * per_row)   # forces chunks of 2 rows
    chunked = resample(x, 30000, 10007)
    assert torch.equal(whole, chunked)
    assert torch.equal(resample(x[3], 30000, 10007), whole[3])       # batch invariance, bit for bit


def test_full_size_clip_fourier_44k1_to_16k(gpu):
    """30 s at 44.1 kHz -> 16 kHz by the Fourier method (the conversion the reference's
    front doors make before the Whisper front-end): two rows against SciPy, eight rows
    through linearity."""
    import scipy.signal as ss

    n, m = 44100 * 30, 16000 * 30
    g = torch.Generator(device=gpu)
    g.manual_seed(11)
    x = torch.randn((8, n), generator=g, device=gpu, dtype=torch.float32)
    y = resample(x, 44100, 16000)
    assert y.shape == (8, m)
    xc = x[[0, 7]].cpu().numpy()
    truth = ss.resample(xc.astype(np.float64), m, axis=-1)
    assert np.abs(y[[0, 7]].cpu().numpy() - truth).max() <= FFT_TOL * np.abs(truth).max()
    mix = resample(x[:4] + 2.0 * x[4:], 44100, 16000)
    assert (mix - (y[:4] + 2.0 * y[4:])).abs().max().item() < 2e-5


def test_resample_golden_vectors(gpu):
    import os

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "resample_golden.npz"))
    x = _dev(gold["x"], gpu)
    for key in gold.files:
        if key == "x":
            continue
        kind, *args = key.split("_")
        if kind == "poly":
            pad = args[2] if len(args) > 2 else "constant"
            got = resample_poly(x, int(args[0]), int(args[1]), padtype=pad).cpu().numpy()
            np.testing.assert_array_equal(got, gold[key], err_msg=key)
        elif kind == "linear":
            got = resample(x, int(args[0]), int(args[1]), res_type="linear").cpu().numpy()
            np.testing.assert_array_equal(got, gold[key], err_msg=key)
        else:
            if len(args) > 2:
                got = resample(x[:, :4409], int(args[0]), int(args[1]), fix=False, scale=True)
            else:
                got = resample(x, int(args[0]), int(args[1]))
            got = got.cpu().numpy()
            assert got.shape == gold[key].shape, key
            assert np.abs(got - gold[key]).max() <= FFT_TOL * np.abs(gold[key]).max(), key


def test_c_abi_argument_checks(gpu):
    lib = _lib.lib()
    x = torch.zeros(8, device=gpu)
    out = torch.zeros(8, device=gpu)
    tab = torch.zeros(4, device=gpu)
    s = torch.cuda.current_stream(gpu).cuda_stream
    assert lib.mia_upfirdn(x.data_ptr(), 1, 8, tab.data_ptr(), 0, 1, 4, 0, 8, 0,
                           out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_upfirdn(x.data_ptr(), 1, 8, tab.data_ptr(), 1, 1, 4, 0, 8, 9,
                           out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert b"padding mode" in lib.mia_last_error()
    assert lib.mia_upfirdn(x.data_ptr(), 1, 8, tab.data_ptr(), 1, 1, 20000, 0, 8, 0,
                           out.data_ptr(), s) == _lib.MIA_ERR_UNSUPPORTED
    assert lib.mia_upfirdn(x.data_ptr(), 0, 8, tab.data_ptr(), 1, 1, 4, 0, 8, 0,
                           out.data_ptr(), s) == _lib.MIA_OK    # empty batch: nothing to do
    assert lib.mia_resample_linear(x.data_ptr(), 1, 8, 4, 5.0, 1.0, out.data_ptr(),
                                   s) == _lib.MIA_ERR_INVALID   # would read past the row
    assert lib.mia_resample_linear(x.data_ptr(), 1, 8, 0, 0.0, 1.0, out.data_ptr(), s) == _lib.MIA_OK
    # Fourier method: sizes, workspace.  Odd lengths take the chirp-z path (tables: both chirps
    # and both filter spectra; workspace: two buffers of the longer convolution), even lengths
    # with 2-3-5-7-smooth halves the direct path (no tables, two half-length buffers).
    assert lib.mia_resample_fft_table_bytes(9, 5) == 8 * (9 + 5 + 32 + 16)
    assert lib.mia_resample_fft_workspace_bytes(3, 9, 5) == 2 * 3 * 32 * 8
    assert lib.mia_resample_fft_table_bytes(8, 4) == 8
    assert lib.mia_resample_fft_workspace_bytes(3, 8, 4) == 2 * 3 * 4 * 8
    assert lib.mia_resample_fft_table_bytes(1 << 28, 4) == 0
    x9 = torch.arange(9, device=gpu, dtype=torch.float32)
    for n_in, n_out in ((9, 5), (8, 4)):
        tab = torch.zeros(lib.mia_resample_fft_table_bytes(n_in, n_out), dtype=torch.uint8, device=gpu)
        work = torch.zeros(lib.mia_resample_fft_workspace_bytes(1, n_in, n_out), dtype=torch.uint8,
                           device=gpu)
        assert lib.mia_resample_fft_tables(n_in, n_out, tab.data_ptr(), work.data_ptr(),
                                           work.numel(), s) == _lib.MIA_OK
        assert lib.mia_resample_fft(x9.data_ptr(), 1, n_in, n_out, 1.0, tab.data_ptr(),
                                    out.data_ptr(), work.data_ptr(), work.numel() - 1,
                                    s) == _lib.MIA_ERR_INVALID
        assert b"workspace too small" in lib.mia_last_error()
        assert lib.mia_resample_fft(x9.data_ptr(), 1, n_in, n_out, 1.0, tab.data_ptr(),
                                    out.data_ptr(), work.data_ptr(), work.numel(), s) == _lib.MIA_OK
        torch.cuda.synchronize()
        want = orc.resample(x9[:n_in].cpu().numpy(), n_in, n_out)
        np.testing.assert_allclose(out[:n_out].cpu().numpy(), want, rtol=0, atol=1e-5)
    assert lib.mia_resample_fft(x9.data_ptr(), 1, 0, 4, 1.0, tab.data_ptr(), out.data_ptr(),
                                work.data_ptr(), work.numel(), s) == _lib.MIA_ERR_INVALID


def test_fourier_tables_built_on_one_stream_are_safe_on_another(gpu):
    """The chirp-z tables are filled asynchronously on the stream of the first call; a second
    call on another stream must wait for them (it used to read them half-built)."""
    import importlib

    rs = importlib.import_module("mlx_audio_amd.primitives.resample")  # (the package re-exports
    # a function of the same name)
    rng = np.random.default_rng(31)
    n, sr_in, sr_out = 30011, 30011, 11003  # a prime length: the chirp-z path, large tables
    x = rng.standard_normal((3, n)).astype(np.float32)
    want = orc.resample(x, sr_in, sr_out)
    xd = torch.from_numpy(x).to(gpu)
    with rs._fft_tables_lock:
        rs._fft_tables_cache.clear()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        # a long kernel queue in front of the table build, so that it is still running when
        # the other stream arrives
        junk = torch.randn((1 << 26,), device=gpu)
        for _ in range(20):
            junk = junk * 1.0001 + 0.5
        ya = resample(xd, sr_in, sr_out)
    with torch.cuda.stream(b):
        yb = resample(xd, sr_in, sr_out)  # cache hit on another stream
    torch.cuda.synchronize()
    peak = np.abs(want).max()
    for got in (ya, yb):
        assert np.abs(got.cpu().numpy() - want).max() <= 2e-6 * peak
    assert torch.equal(ya, yb)


def test_fourier_tables_evicted_while_another_stream_still_reads_them(gpu):
    """The table cache keeps eight entries.  A stream that got its tables by a cache hit may
    still have kernels queued on them when later calls (on the building stream) evict the
    entry and the allocator reuses the block: the hit registers its stream with the block, so
    the reuse waits.  The queued call's result must be what a quiet run gives."""
    import importlib

    rs = importlib.import_module("mlx_audio_amd.primitives.resample")
    rng = np.random.default_rng(32)
    n, sr_out = 30011, 11003
    x = rng.standard_normal((3, n)).astype(np.float32)
    xd = torch.from_numpy(x).to(gpu)
    with rs._fft_tables_lock:
        rs._fft_tables_cache.clear()
    quiet = resample(xd, n, sr_out).clone()
    torch.cuda.synchronize()
    b = torch.cuda.Stream()
    with torch.cuda.stream(b):
        junk = torch.randn((1 << 26,), device=gpu)
        for _ in range(20):
            junk = junk * 1.0001 + 0.5
        yb = resample(xd, n, sr_out)          # cache hit on b, queued behind the junk
    # meanwhile, on the building stream: nine other lengths evict the entry and allocate anew
    others = [resample(xd[:, :n - 2 * k], n - 2 * k, sr_out) for k in range(1, 10)]
    assert (n, quiet.shape[-1], xd.device.index) not in rs._fft_tables_cache
    torch.cuda.synchronize()
    assert torch.equal(yb, quiet)
    assert all(torch.isfinite(o).all() for o in others)
