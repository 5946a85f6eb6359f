"""Griffin-Lim on the GPU against the oracle.  A single iteration is compared tightly; runs of
several iterations start from the same seeded phase and are compared with a tolerance that
grows with the iteration count (each iteration re-normalises phases, which amplifies float32
rounding where a bin is nearly empty); long runs are judged by what the algorithm is for: the
spectrogram of the result approaches the target as fast as the oracle's does."""

import numpy as np
import pytest
import torch

import oracle.griffinlim as orc
import oracle.spectral as sp
from mlx_audio_amd import _lib
from mlx_audio_amd.primitives import griffinlim, griffinlim_iter, magnitude, stft

pytestmark = pytest.mark.gpu


def _dev(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def _target(n=8192, n_fft=512, hop=128, rows=2, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    y = np.stack([np.sin(2 * np.pi * (300 + 200 * r) * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t))
                  + 0.05 * rng.standard_normal(n) for r in range(rows)]).astype(np.float32)
    return y, np.abs(sp.stft(y, n_fft=n_fft, hop_length=hop)).astype(np.float32)


def test_single_iteration_matches(gpu):
    _, S = _target()
    rng = np.random.default_rng(1)
    ang = rng.uniform(-np.pi, np.pi, S.shape).astype(np.float32)
    prev = (S * np.exp(1j * rng.uniform(-np.pi, np.pi, S.shape))).astype(np.complex64)
    for tprev, mom in ((None, 0.99), (prev, 0.99), (prev, 0.0)):
        want_ang, want_out, want_err = orc.griffinlim_iter(S, ang, 128, 512, 512, momentum=mom, tprev=tprev)
        got_ang, got_out, got_err = griffinlim_iter(_dev(S, gpu), _dev(ang, gpu), 128, 512, 512, momentum=mom,
                                                    tprev=None if tprev is None else _dev(tprev, gpu))
        scale = np.abs(want_out).max()
        assert np.abs(got_out.cpu().numpy() - want_out).max() <= 2e-4 * scale
        np.testing.assert_allclose(got_err.item(), want_err, rtol=1e-4)
        # phases agree wherever the bin is not nearly empty (modulo 2 pi)
        strong = S > 1e-2 * S.max()
        d = np.angle(np.exp(1j * (got_ang.cpu().numpy() - want_ang)))
        assert np.abs(d[strong]).max() < 1e-3


@pytest.mark.parametrize("kw", [{}, {"momentum": 0.0}, {"init": "zeros"}, {"length": 8000},
                                {"hop_length": 64, "win_length": 400, "window": "hamming"}])
def test_short_runs_track_the_oracle(gpu, kw):
    y, S = _target()
    if "hop_length" in kw:
        S = np.abs(sp.stft(y, n_fft=512, **kw)).astype(np.float32)
    for n_iter, tol in ((1, 2e-4), (3, 2e-3)):
        want = orc.griffinlim(S, n_iter=n_iter, random_state=7, **kw)
        got = griffinlim(_dev(S, gpu), n_iter=n_iter, random_state=7, **kw).cpu().numpy()
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= tol * np.abs(want).max(), (n_iter, kw)
    one = griffinlim(_dev(S[0], gpu), n_iter=1, random_state=7, **kw)
    assert one.dim() == 1


def test_in_place_iteration_equals_the_trimmed_form(gpu, monkeypatch):
    """Feeding the forward transform from the untrimmed overlap-add buffer gives bit for bit
    what istft() followed by stft() gives."""
    import importlib

    mod = importlib.import_module("mlx_audio_amd.primitives.griffinlim")
    y, S = _target()
    cases = [({}, S), ({"length": 8000}, S), ({"momentum": 0.0, "init": "zeros"}, S)]
    kw = {"hop_length": 64, "win_length": 400, "window": "hamming"}
    cases.append((kw, np.abs(sp.stft(y, n_fft=512, **kw)).astype(np.float32)))
    for kw, Sk in cases:
        fast = griffinlim(_dev(Sk, gpu), n_iter=3, random_state=1, **kw)
        monkeypatch.setattr(mod, "_FEED_UNTRIMMED", False)
        slow = griffinlim(_dev(Sk, gpu), n_iter=3, random_state=1, **kw)
        monkeypatch.setattr(mod, "_FEED_UNTRIMMED", True)
        assert torch.equal(fast, slow), kw


def test_long_run_converges_like_the_oracle(gpu):
    y, S = _target(n=16384, n_fft=1024, hop=256)
    got = griffinlim(_dev(S, gpu), n_iter=32, random_state=0).cpu().numpy()
    want = orc.griffinlim(S, n_iter=32, random_state=0)

    def consistency(x):
        M = np.abs(sp.stft(x, n_fft=1024, hop_length=256))
        return np.linalg.norm(M - S) / np.linalg.norm(S)

    cg, cw = consistency(got), consistency(want)
    assert cg < 0.12 and abs(cg - cw) < 0.02, (cg, cw)
    assert consistency(griffinlim(_dev(S, gpu), n_iter=2, random_state=0).cpu().numpy()) > cg


def test_full_size_batch(gpu):
    """64 x 10 s @22.05 kHz, n_fft 2048 / hop 512 (the fused inverse kernel's framing): eight
    iterations; the magnitude of the result is closer to the target than after one."""
    g = torch.Generator(device=gpu)
    g.manual_seed(1)
    y = torch.randn((64, 220500), generator=g, device=gpu)
    S = magnitude(stft(y, n_fft=2048, hop_length=512))
    def err(n_iter):
        r = griffinlim(S, n_iter=n_iter, random_state=0)
        return ((magnitude(stft(r, n_fft=2048, hop_length=512)) - S).norm() / S.norm()).item()
    e1, e8 = err(1), err(8)
    assert e8 < e1 and e8 < 0.35, (e1, e8)


def test_c_abi_checks(gpu):
    lib = _lib.lib()
    z = torch.zeros(8, dtype=torch.complex64, device=gpu)
    m = torch.ones(8, device=gpu)
    s = torch.cuda.current_stream(gpu).cuda_stream
    assert lib.mia_griffinlim_update(z.data_ptr(), m.data_ptr(), z.data_ptr(), z.data_ptr(), 8, 1.0, s) == _lib.MIA_ERR_INVALID
    out = torch.empty(8, dtype=torch.complex64, device=gpu)
    prev = torch.zeros(8, dtype=torch.complex64, device=gpu)
    assert lib.mia_griffinlim_update(z.data_ptr(), m.data_ptr(), prev.data_ptr(), out.data_ptr(), 8, 0.5, s) == _lib.MIA_OK
    torch.cuda.synchronize()
    # a zero spectrum has phase 0: r = mag, rebuilt = 1.5 * mag
    np.testing.assert_allclose(out.cpu().numpy(), 1.5 + 0j)
    np.testing.assert_allclose(prev.cpu().numpy(), 1.0 + 0j)
    assert lib.mia_polar(m.data_ptr(), m.data_ptr(), 8, out.data_ptr(), s) == _lib.MIA_OK
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), np.exp(1j), rtol=1e-6)


@pytest.mark.parametrize("shape", [(2, 257, 65), (1, 33, 1), (3, 5, 130), (1, 1025, 44)])
def test_set_up_in_one_pass(gpu, shape):
    """mia_griffinlim_init: transposed magnitudes and S * exp(i angle) (twice: estimate and previous
    estimate) in the loop's (rows, frames, bins) order, against mia_polar on transposed copies."""
    from mlx_audio_amd import _lib
    from mlx_audio_amd.primitives.griffinlim import _polar

    rows, bins, frames = shape
    rng = np.random.default_rng(bins)
    S = _dev(np.abs(rng.standard_normal(shape)).astype(np.float32), gpu)
    ang = _dev(rng.uniform(-np.pi, np.pi, shape).astype(np.float32), gpu)
    lib = _lib.lib()
    s = torch.cuda.current_stream(gpu).cuda_stream
    for angles in (ang, None):
        mag_t = torch.full((rows, frames, bins), -1.0, device=gpu)
        reb = torch.full((rows, frames, bins), complex(9, 9), dtype=torch.complex64, device=gpu)
        prev = torch.full_like(reb, complex(8, 8))
        _lib.check(lib.mia_griffinlim_init(S.data_ptr(), 0 if angles is None else angles.data_ptr(), rows, bins,
                                           frames, mag_t.data_ptr(), reb.data_ptr(), prev.data_ptr(), s))
        want_mag = S.transpose(1, 2).contiguous()
        want = _polar(want_mag, (ang if angles is not None else torch.zeros_like(ang)).transpose(1, 2).contiguous())
        assert torch.equal(mag_t, want_mag)
        assert torch.equal(torch.view_as_real(reb), torch.view_as_real(want))
        assert torch.equal(torch.view_as_real(prev), torch.view_as_real(want))
    assert lib.mia_griffinlim_init(0, 0, rows, bins, frames, mag_t.data_ptr(), reb.data_ptr(), prev.data_ptr(), s) \
        == _lib.MIA_ERR_INVALID

