"""CPU-side checks of the C ABI (include/mia.h -> libmia_hip.so).

No compute entry point is called here (there is no GPU on the CPU runner):
the library must load, export every symbol the header declares, and its
host-side table builders must reproduce the oracle bit for bit.
"""

import ctypes
import os
import re

import numpy as np
import pytest

import oracle.spectral as o
from mlx_audio_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mia.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mia_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_and_loaded():
    assert os.path.exists(_lib.LIB_PATH), "build with python mlx_audio_amd/csrc/build.py"
    assert _lib.HAS_HIP_EXT and _lib.has_hip_extensions()
    assert _lib.lib().mia_version() >= 1


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared_symbols()
    assert len(declared) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} is declared in mia.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"


def test_header_has_no_torch_or_cpp_types():
    text = open(HEADER).read()
    assert 'extern "C"' in text
    for banned in ("torch", "at::", "std::", "hipStream_t"):
        assert banned not in re.sub(r"/\*.*?\*/", "", text, flags=re.S)


@pytest.mark.parametrize("name", ["hann", "hanning", "hamming", "blackman", "bartlett",
                                  "triangular", "rectangular", "boxcar", "ones"])
@pytest.mark.parametrize("n", [1, 2, 4, 255, 400, 1024, 4096])
@pytest.mark.parametrize("periodic", [1, 0])
def test_window_builder_bit_exact(name, n, periodic):
    buf = np.empty(n, dtype=np.float32)
    rc = _lib.lib().mia_generate_window(name.encode(), n, periodic,
                                        buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == _lib.MIA_OK
    np.testing.assert_array_equal(buf, o.get_window(name, n, bool(periodic)))


def test_window_builder_errors():
    buf = (ctypes.c_float * 4)()
    lib = _lib.lib()
    assert lib.mia_generate_window(b"kaiser", 4, 1, buf) == _lib.MIA_ERR_INVALID
    assert b"kaiser" in lib.mia_last_error()
    assert lib.mia_generate_window(b"hann", 0, 1, buf) == _lib.MIA_ERR_INVALID
    with pytest.raises(ValueError):
        _lib.check(lib.mia_generate_window(b"nope", 4, 1, buf))


@pytest.mark.parametrize("htk", [0, 1])
def test_hz_mel_bit_exact(htk):
    f = np.array([0.0, 1.0, 55.0, 440.0, 999.9, 1000.0, 1000.1, 4000.0, 8000.0, 22050.0])
    dptr = ctypes.POINTER(ctypes.c_double)
    out = np.empty_like(f)
    assert _lib.lib().mia_hz_to_mel(f.ctypes.data_as(dptr), len(f), htk,
                                    out.ctypes.data_as(dptr)) == 0
    np.testing.assert_allclose(out, o.hz_to_mel(f, bool(htk)), rtol=1e-15, atol=0)
    back = np.empty_like(f)
    assert _lib.lib().mia_mel_to_hz(out.ctypes.data_as(dptr), len(f), htk,
                                    back.ctypes.data_as(dptr)) == 0
    np.testing.assert_allclose(back, o.mel_to_hz(out, bool(htk)), rtol=1e-15, atol=0)
    np.testing.assert_allclose(back, f, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax,htk,norm", [
    (22050, 2048, 128, 0.0, None, False, "slaney"), (16000, 400, 80, 0.0, 8000.0, False, "slaney"),
    (16000, 400, 128, 0.0, 8000.0, False, "slaney"), (44100, 4096, 128, 0.0, None, False, "slaney"),
    (8000, 256, 40, 0.0, None, False, "slaney"), (48000, 1024, 64, 50.0, 14000.0, True, "slaney"),
    (16000, 2048, 128, 0.0, None, False, "slaney"), (16000, 512, 40, 20.0, 7600.0, True, None)])
def test_mel_filterbank_builder_bit_exact(sr, n_fft, n_mels, fmin, fmax, htk, norm):
    out = np.empty((n_mels, n_fft // 2 + 1), dtype=np.float32)
    rc = _lib.lib().mia_mel_filterbank(sr, n_fft, n_mels, fmin, sr / 2 if fmax is None else fmax,
                                       int(htk), 0 if norm is None else 1,
                                       out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    np.testing.assert_array_equal(out, o.mel_filterbank(sr, n_fft, n_mels, fmin, fmax, htk, norm))


def test_mel_filterbank_builder_errors():
    out = (ctypes.c_float * (80 * 201))()
    lib = _lib.lib()
    assert lib.mia_mel_filterbank(16000, 400, 0, 0.0, 8000.0, 0, 1, out) == _lib.MIA_ERR_INVALID
    assert lib.mia_mel_filterbank(16000, 400, 80, -1.0, 8000.0, 0, 1, out) == _lib.MIA_ERR_INVALID
    assert lib.mia_mel_filterbank(16000, 400, 80, 0.0, 9000.0, 0, 1, out) == _lib.MIA_ERR_INVALID
    assert lib.mia_mel_filterbank(16000, 400, 80, 5000.0, 4000.0, 0, 1, out) == _lib.MIA_ERR_INVALID


@pytest.mark.parametrize("n_out,n_in", [(40, 128), (13, 40), (20, 128), (64, 64), (1, 7)])
@pytest.mark.parametrize("ortho", [1, 0])
def test_dct_matrix_builder_bit_exact(n_out, n_in, ortho):
    out = np.empty((n_out, n_in), dtype=np.float32)
    assert _lib.lib().mia_dct_matrix(n_out, n_in, ortho,
                                     out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))) == 0
    want = o.dct_matrix(n_out, n_in, "ortho" if ortho else None)
    # cos() of the C library and of NumPy may differ in the last float64 bit,
    # which can move a float32 rounding: allow one float32 ulp
    np.testing.assert_allclose(out, want, rtol=0, atol=6e-8)


def test_supported_fft_sizes():
    lib = _lib.lib()
    for n in (256, 400, 512, 800, 1024, 2048, 4096):
        assert lib.mia_fft_size_supported(n) == 1
    for n in (0, 100, 300, 8192):
        assert lib.mia_fft_size_supported(n) == 0


def test_null_plan_is_rejected_without_touching_a_gpu():
    lib = _lib.lib()
    assert lib.mia_plan_num_frames(None, 1000) < 0
    assert lib.mia_plan_destroy(None) in (_lib.MIA_OK, _lib.MIA_ERR_INVALID)
    assert lib.mia_stft(None, None, 1, 1, 1.0, None, None) == _lib.MIA_ERR_INVALID
    assert lib.mia_melspectrogram(None, None, 1, 1, 0, 1.0, 1e-10, 10.0, 80.0, 0, None, None, 0,
                                  None) == _lib.MIA_ERR_INVALID


def test_compiled_whisper_banks_are_current_and_bit_exact():
    """The generated sweep of the n_fft = 400 kernel is built from the same filterbank
    arithmetic as mia_mel_filterbank, and the file on disk is what the generator writes."""
    import importlib.util
    import os

    from mlx_audio_amd.primitives.mel import mel_filterbank_np

    path = os.path.join(os.path.dirname(__file__), "..", "mlx_audio_amd", "csrc",
                        "gen_w400_sweep.py")
    spec = importlib.util.spec_from_file_location("gen_w400_sweep", path)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    for n_mels in gen.BANKS:
        want = mel_filterbank_np(16000, 400, n_mels, 0.0, 8000.0, False, "slaney")
        got = gen.filterbank(n_mels)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert gen.generate() == open(gen.OUT).read()


def test_upfirdn_margin_is_host_only_and_says_where_the_shared_window_kernel_applies():
    """mia_upfirdn_margin touches no device: zero for rows shorter than one run of outputs and
    for ratios whose lanes' samples do not fit the LDS, else the number of zeros
    mia_upfirdn_shared wants on both sides of every phase-table row; it grows with down / up
    (the offset between the windows of adjacent outputs)."""
    lib = _lib.lib()
    long_row = 10 ** 7
    assert lib.mia_upfirdn_margin(160, 441, 56, 1000) == 0
    assert lib.mia_upfirdn_margin(1, 8, 161, long_row) == 0     # rows too long for a lane each
    assert lib.mia_upfirdn_margin(147, 160, 22, long_row) == 0  # lanes a multiple of 4 samples
    assert lib.mia_upfirdn_margin(441, 160, 21, long_row) == 0  # apart and a short filter
    assert lib.mia_upfirdn_margin(0, 3, 61, long_row) == 0      # not a ratio
    margins = [lib.mia_upfirdn_margin(1, d, 20 * d + 1, long_row) for d in (1, 2, 3, 5, 6, 7)]
    assert all(m > 0 for m in margins) and margins == sorted(margins)
    for up, down, taps in ((160, 441, 56), (3, 1, 21), (2, 1, 21), (160, 147, 21), (320, 441, 28),
                           (1, 4, 81), (3, 8, 55)):
        assert lib.mia_upfirdn_margin(up, down, taps, long_row) > 0, (up, down)


def test_shared_window_plans_fit_the_lds_for_every_ratio():
    """The shared-window kernel's LDS holds the staged samples AND a table of step constants
    that grows with up: whatever ratio mia_upfirdn_margin accepts must fit a CU's 160 KiB
    (integer upsampling by 32, 48, 147 and 100/3 once did not)."""
    lib = _lib.lib()
    long_row = 10 ** 7
    worst = 0
    for up in list(range(1, 65)) + [80, 96, 100, 147, 160, 200, 320, 441, 640, 1000]:
        for down in (1, 2, 3, 5, 7, 8, 147, 160, 441):
            taps = 20 * max(up, down) // up + 2
            lds = lib.mia_upfirdn_shared_lds_bytes(up, down, taps)
            assert (lds > 0) == (lib.mia_upfirdn_margin(up, down, taps, long_row) > 0), (up, down)
            assert lds <= 160 * 1024, (up, down, lds)
            worst = max(worst, lds)
    assert worst > 64 * 1024   # the sweep does reach plans with large spans
    for up, down in ((32, 1), (48, 1), (147, 1), (100, 3)):
        assert 0 < lib.mia_upfirdn_shared_lds_bytes(up, down, 22) <= 160 * 1024

