"""Host-side behaviour of the Python mirror that needs no GPU: argument
validation (same exception types and messages as the reference raises, see
primitives/stft.py:190-199, mel.py:227-237, mfcc.py:226-231, convert.py:49-52),
host table helpers, the plan-cache key, and the refusal to run without a GPU.
"""

import os

import numpy as np
import pytest
import torch

import oracle.spectral as o
from mlx_audio_amd import _lib, _runtime, primitives
from mlx_audio_amd.primitives import windows as win_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NO_GPU = not torch.cuda.is_available()


def test_stft_argument_validation_precedes_device_work():
    x = np.zeros(4096, dtype=np.float32)
    with pytest.raises(ValueError, match="hop_length must be positive"):
        primitives.stft(x, hop_length=0)
    with pytest.raises(ValueError, match="win_length must be positive"):
        primitives.stft(x, win_length=0)
    with pytest.raises(ValueError, match=r"win_length \(4096\) must be <= n_fft \(2048\)"):
        primitives.stft(x, win_length=4096)
    with pytest.raises(ValueError, match="should typically be <= n_fft"):
        primitives.stft(x, hop_length=4096)
    with pytest.raises(ValueError, match="Unknown pad_mode"):
        primitives.stft(x, pad_mode="wrap")


def test_mfcc_and_db_validation():
    x = np.zeros(4096, dtype=np.float32)
    with pytest.raises(ValueError, match="n_mfcc must be positive"):
        primitives.mfcc(x, n_mfcc=0)
    with pytest.raises(ValueError, match="Only DCT type 2"):
        primitives.mfcc(x, dct_type=3)
    with pytest.raises(ValueError, match="Only DCT type 2"):
        primitives.dct(x, type=1)


def test_mel_filterbank_validation_and_values():
    with pytest.raises(ValueError, match="n_mels must be positive"):
        primitives.mel_filterbank(16000, 400, n_mels=0)
    with pytest.raises(ValueError, match="fmin must be non-negative"):
        primitives.mel_filterbank(16000, 400, fmin=-1.0)
    with pytest.raises(ValueError, match="must be less than fmax"):
        primitives.mel_filterbank(16000, 400, fmin=5000.0, fmax=4000.0)
    with pytest.raises(ValueError, match="cannot exceed Nyquist"):
        primitives.mel_filterbank(16000, 400, fmax=9000.0)
    with pytest.raises(ValueError, match="Unknown norm"):
        primitives.mel_filterbank(16000, 400, norm="l2")
    fb = primitives.mel_filterbank(16000, 400, 80, fmax=8000.0, device="cpu")
    assert fb.dtype == torch.float32 and tuple(fb.shape) == (80, 201)
    np.testing.assert_array_equal(fb.numpy(), o.mel_filterbank(16000, 400, 80, 0.0, 8000.0))
    np.testing.assert_allclose(primitives.hz_to_mel(np.array([1000.0])), [15.0])
    np.testing.assert_allclose(primitives.mel_to_hz(np.array([15.0])), [1000.0])


def test_window_helpers():
    np.testing.assert_array_equal(win_mod.window_np("hann", 400), o.get_window("hann", 400))
    np.testing.assert_array_equal(win_mod.padded_window_np("hamming", 300, 512),
                                  o.padded_window("hamming", 300, 512))
    with pytest.raises(ValueError):
        win_mod.window_np("kaiser", 16)
    with pytest.raises(ValueError):
        win_mod.window_np(np.ones(8, dtype=np.float32), 16)
    with pytest.raises(TypeError):
        win_mod.window_np(1.5, 16)
    w = primitives.get_window("blackman", 64, device="cpu")
    np.testing.assert_array_equal(w.numpy(), o.get_window("blackman", 64))
    assert primitives.check_nola("hann", 512, 2048)


def test_status_codes_map_to_reference_exception_types():
    lib = _lib.lib()
    buf = np.empty(4, dtype=np.float32)
    import ctypes
    rc = lib.mia_generate_window(b"bogus", 4, 1, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    with pytest.raises(ValueError, match="Unknown window type"):
        _lib.check(rc)
    _lib.check(_lib.MIA_OK)
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.MIA_ERR_UNSUPPORTED)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.MIA_ERR_HIP)


def test_plan_cache_key_covers_every_table_input():
    a = _runtime._digest(np.arange(4, dtype=np.float32))
    b = _runtime._digest(np.arange(4, dtype=np.float32) + 1)
    assert a != b and _runtime._digest(None) == "-"
    with pytest.raises(ValueError, match="Unknown pad_mode"):
        _runtime.get_plan(torch.device("cuda", 0), 400, 160, True, "wrap", np.ones(400, np.float32))


@pytest.mark.skipif(not NO_GPU, reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback():
    x = np.zeros(4096, dtype=np.float32)
    for call in (lambda: primitives.stft(x), lambda: primitives.melspectrogram(x),
                 lambda: primitives.mfcc(x), lambda: primitives.istft(np.zeros((1025, 4), np.complex64)),
                 lambda: primitives.power_to_db(x), lambda: primitives.dct(x)):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()


def test_product_never_imports_the_oracle():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "mlx_audio_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_traffic_is_reported_only_for_unchanged_kernel_sources(tmp_path):
    """bench.py's roofline.traffic comes from committed counter passes; an entry counts only
    while every source file it names still has the blob hash recorded with it."""
    import json
    import subprocess

    from mlx_audio_amd import provenance as pv

    src = tmp_path / "mlx_audio_amd" / "csrc"
    src.mkdir(parents=True)
    (tmp_path / "profiles").mkdir()
    f = src / "onset.hip"
    f.write_text("__global__ void k() {}\n")
    # the hash is git's own
    want = subprocess.run(["git", "hash-object", str(f)], capture_output=True, text=True).stdout.strip()
    assert pv.blob_hash(str(f)) == want
    hashes = pv.source_hashes("onset", root=str(tmp_path))
    assert hashes == {"mlx_audio_amd/csrc/onset.hip": want}
    table = {"onset:4096": {"hbm_bytes_per_step": 123, "kernel_sources": hashes, "profile": "p.txt"},
             "contrast:512": {"hbm_bytes_per_step": 5}}
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps(table))
    got, info = pv.measured_traffic("onset", 4096, root=str(tmp_path))
    assert got == 123 and info["status"].startswith("sources match") and info["profile"] == "p.txt"
    assert pv.measured_traffic("onset", 512, root=str(tmp_path))[0] is None        # no pass at that size
    got, info = pv.measured_traffic("contrast", 512, root=str(tmp_path))          # no hashes recorded
    assert got is None and "no source hashes" in info["status"]
    f.write_text("__global__ void k() { }\n")                                     # the kernel changed
    got, info = pv.measured_traffic("onset", 4096, root=str(tmp_path))
    assert got is None and info["status"].startswith("stale") and "onset.hip" in info["status"]
    # every bench workload with a counter entry names its sources
    committed = json.load(open(os.path.join(ROOT, pv.TRAFFIC_FILE)))
    for key in committed:
        if not key.startswith("_"):
            assert key.split(":")[0] in pv.WORKLOAD_SOURCES, key
    for names in pv.WORKLOAD_SOURCES.values():
        for name in names:
            assert os.path.exists(os.path.join(ROOT, pv.CSRC, name)), name
