"""HTDemucs input normalisation (reference models/demucs/model.py:206-233) on the GPU against
the NumPy restatement.  Tolerance 2e-5: the statistics are accumulated in double here and in
float32 (pairwise) there."""

import numpy as np
import pytest
import torch

import oracle.spectral as o
from mlx_audio_amd import _lib
from mlx_audio_amd.frontends import htdemucs_normalize, htdemucs_spec

pytestmark = pytest.mark.gpu


def _dev(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


@pytest.mark.parametrize("B,C,T", [(1, 2, 44100), (3, 2, 30000), (2, 1, 5000)])
def test_normalisation_of_mix_and_spectrogram(gpu, B, C, T):
    rng = np.random.default_rng(B * 10 + C)
    mix = (0.1 * rng.standard_normal((B, C, T)) + 0.02).astype(np.float32)
    mix *= np.arange(1, B + 1, dtype=np.float32)[:, None, None]      # different scale per item
    md = _dev(mix, gpu)
    spec = htdemucs_spec(md)                                          # strided view, as the model passes it
    got = htdemucs_normalize(md, spec)
    want = o.htdemucs_normalize(mix, spec.cpu().numpy())
    for g, w, name in zip(got, want, ("freq_in", "spec_mean", "spec_std", "mix_norm", "mix_mean", "mix_std")):
        assert tuple(g.shape) == w.shape, name
        np.testing.assert_allclose(g.cpu().numpy(), w, rtol=2e-5, atol=2e-5, err_msg=name)
    freq_in = got[0]
    assert freq_in.shape == (B, 2 * C, 2048, int(np.ceil(T / 1024)))
    # each item ends up with zero mean and unit deviation
    flat = freq_in.reshape(B, -1)
    assert flat.mean(dim=1).abs().max().item() < 1e-4
    assert (flat.std(dim=1, unbiased=False) - 1).abs().max().item() < 1e-3
    # the default computes the spectrogram itself, with the sums added up by the STFT kernel
    # (another order of double-precision additions: equal to float rounding of mean and std)
    again = htdemucs_normalize(md)
    for a, g in zip(again, got):
        np.testing.assert_allclose(a.cpu().numpy(), g.cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_stft_with_moments(gpu):
    """mia_stft_moments: the spectrum equals mia_stft's bit for bit, the sums equal a double-
    precision sum over the kept bins and frames, per group of rows."""
    from mlx_audio_amd._runtime import get_plan, stream_ptr
    from mlx_audio_amd.primitives.windows import padded_window_np

    lib = _lib.lib()
    rng = np.random.default_rng(3)
    s = stream_ptr(gpu)
    for n_fft, hop, rows, group, T in ((4096, 1024, 4, 2, 50000), (1024, 256, 6, 3, 9000), (512, 128, 2, 1, 3000)):
        y = _dev((rng.standard_normal((rows, T)) * (1 + np.arange(rows))[:, None]).astype(np.float32), gpu)
        plan = get_plan(gpu, n_fft, hop, True, "reflect", padded_window_np("hann", n_fft, n_fft))
        F = plan.num_frames(T)
        shape = (rows, F, n_fft // 2 + 1)
        ref = torch.empty(shape, dtype=torch.complex64, device=gpu)
        out = torch.empty(shape, dtype=torch.complex64, device=gpu)
        sums = torch.full((rows // group, 2), 7.0, dtype=torch.float64, device=gpu)   # the call zeroes them
        _lib.check(lib.mia_stft(plan.handle, y.data_ptr(), rows, T, 0.25, ref.data_ptr(), s))
        lo, hi, nb = 2, F - 1, n_fft // 2
        _lib.check(lib.mia_stft_moments(plan.handle, y.data_ptr(), rows, T, 0.25, out.data_ptr(), group, nb,
                                        lo, hi, sums.data_ptr(), s))
        assert torch.equal(torch.view_as_real(out), torch.view_as_real(ref))
        kept = torch.view_as_real(ref[:, lo:hi, :nb]).to(torch.float64).reshape(rows // group, -1)
        np.testing.assert_allclose(sums[:, 0].cpu().numpy(), kept.sum(dim=1).cpu().numpy(), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(sums[:, 1].cpu().numpy(), (kept * kept).sum(dim=1).cpu().numpy(), rtol=1e-11)
    assert lib.mia_stft_moments(plan.handle, y.data_ptr(), 2, 3000, 1.0, out.data_ptr(), 3, 10, 0, 1,
                                sums.data_ptr(), s) == _lib.MIA_ERR_INVALID     # 2 rows, groups of 3
    assert lib.mia_stft_moments(plan.handle, y.data_ptr(), 2, 3000, 1.0, out.data_ptr(), 1, 999, 0, 1,
                                sums.data_ptr(), s) == _lib.MIA_ERR_INVALID     # more bins than the plan has
    odd = get_plan(gpu, 1000, 250, True, "reflect", padded_window_np("hann", 1000, 1000))
    assert lib.mia_stft_moments(odd.handle, y.data_ptr(), 2, 3000, 1.0, out.data_ptr(), 1, 10, 0, 1,
                                sums.data_ptr(), s) == _lib.MIA_ERR_UNSUPPORTED


def test_contiguous_spectrogram_odd_tile_sizes_and_c_abi(gpu):
    rng = np.random.default_rng(5)
    spec = (rng.standard_normal((2, 3, 45, 70)) + 1j * rng.standard_normal((2, 3, 45, 70))).astype(np.complex64)
    mix = rng.standard_normal((2, 3, 777)).astype(np.float32)
    got = htdemucs_normalize(_dev(mix, gpu), _dev(spec, gpu))
    want = o.htdemucs_normalize(mix, spec)
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.cpu().numpy(), w, rtol=2e-5, atol=2e-5)
    # frames fastest with a row pitch of its own (a slice of a larger array)
    big_spec = (rng.standard_normal((2, 3, 50, 90)) + 1j * rng.standard_normal((2, 3, 50, 90))).astype(np.complex64)
    sliced = _dev(big_spec, gpu)[:, :, 3:48, 11:81]
    assert sliced.stride(3) == 1 and sliced.stride(2) == 90
    got = htdemucs_normalize(_dev(mix, gpu), sliced)
    want = o.htdemucs_normalize(mix, big_spec[:, :, 3:48, 11:81])
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.cpu().numpy(), w, rtol=2e-5, atol=2e-5)
    with pytest.raises(ValueError, match="complex64"):
        htdemucs_normalize(_dev(mix, gpu), _dev(spec.real.copy(), gpu))
    lib = _lib.lib()
    x = torch.ones(8, device=gpu)
    sums = torch.zeros(4, dtype=torch.float64, device=gpu)
    s = torch.cuda.current_stream(gpu).cuda_stream
    assert lib.mia_normalize_rows(x.data_ptr(), 1, 0, 1e-5, x.data_ptr(), x.data_ptr(), sums.data_ptr(), s) == _lib.MIA_ERR_INVALID
    # more rows than one launch's grid takes: the entry point goes through them in pieces
    big = torch.randn((70000, 16), device=gpu, generator=torch.Generator(device=gpu).manual_seed(11))
    big_out = torch.empty_like(big)
    big_st = torch.empty((70000, 2), device=gpu)
    big_sums = torch.zeros((70000, 2), dtype=torch.float64, device=gpu)
    assert lib.mia_normalize_rows(big.data_ptr(), 70000, 16, 1e-5, big_out.data_ptr(), big_st.data_ptr(),
                                  big_sums.data_ptr(), s) == _lib.MIA_OK
    torch.cuda.synchronize()
    mean = big.double().mean(dim=1)
    std = big.double().std(dim=1, unbiased=False)
    np.testing.assert_allclose(big_st[:, 0].cpu().numpy(), mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(big_out.cpu().numpy(),
                               ((big.double() - mean[:, None]) / (std[:, None] + 1e-5)).cpu().numpy(),
                               rtol=2e-5, atol=2e-5)
    out = torch.empty(8, device=gpu)
    st = torch.empty(2, device=gpu)
    assert lib.mia_normalize_rows(x.data_ptr(), 1, 8, 1e-5, out.data_ptr(), st.data_ptr(), sums.data_ptr(), s) == _lib.MIA_OK
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), 0.0)               # a constant row: (1 - 1) / (0 + eps)
    np.testing.assert_allclose(st.cpu().numpy(), [1.0, 1e-5], rtol=1e-6)


@pytest.mark.parametrize("rows,F,T,pk,pt", [(3, 45, 70, 1, 2), (1, 32, 64, 0, 0), (2, 1, 1, 1, 2), (2, 2048, 87, 1, 2),
                                            (1, 33, 129, 3, 1)])
def test_planes_to_spectrum(gpu, rows, F, T, pk, pt):
    """mia_cac_to_spectrum: real / imaginary planes -> the (rows, frames, bins) complex spectrum with
    zero bins appended and zero frames on both sides, bit for bit what assignment into a zeroed
    array gives; and htdemucs_ispec through it equals the inverse STFT of that array."""
    rng = np.random.default_rng(rows * 100 + F)
    planes = rng.standard_normal((rows, 2, F, T)).astype(np.float32)
    want = np.zeros((rows, T + 2 * pt, F + pk), dtype=np.complex64)
    want[:, pt:pt + T, :F] = (planes[:, 0] + 1j * planes[:, 1]).transpose(0, 2, 1)
    out = torch.full((rows, T + 2 * pt, F + pk), complex(7, 7), dtype=torch.complex64, device=gpu)
    lib = _lib.lib()
    s = torch.cuda.current_stream(gpu).cuda_stream
    _lib.check(lib.mia_cac_to_spectrum(_dev(planes, gpu).data_ptr(), rows, F, T, pk, pt, out.data_ptr(), s))
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    assert lib.mia_cac_to_spectrum(0, rows, F, T, pk, pt, out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_cac_to_spectrum(out.data_ptr(), rows, 0, T, pk, pt, out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_cac_to_spectrum(0, 0, F, T, pk, pt, 0, s) == _lib.MIA_OK


def test_ispec_equals_the_inverse_stft_of_the_assembled_spectrum(gpu):
    from mlx_audio_amd.frontends import htdemucs_ispec
    from mlx_audio_amd.primitives import istft

    rng = np.random.default_rng(9)
    B, S, C, F, Tf, length = 2, 2, 2, 2048, 12, 9000
    m = rng.standard_normal((B, S, 2 * C, F, Tf)).astype(np.float32)
    got = htdemucs_ispec(_dev(m, gpu), length)
    mr = m.reshape(B * S * C, 2, F, Tf)
    z = np.zeros((B * S * C, F + 1, Tf + 4), dtype=np.complex64)
    z[:, :F, 2:2 + Tf] = mr[:, 0] + 1j * mr[:, 1]
    pad = 1024 // 2 * 3
    le = 1024 * int(np.ceil(length / 1024)) + 2 * pad
    y = istft(_dev(z, gpu), hop_length=1024, n_fft=4096, length=le) * np.float32(np.sqrt(4096))
    want = y.reshape(B, S, C, -1)[:, :, :, pad:pad + length]
    assert got.shape == (B, S, C, length)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=2e-5 * float(want.abs().max()))


@pytest.mark.parametrize("rows,group,n,zl,zr", [(4, 2, 44100, 1536, 1700), (6, 3, 1000, 0, 0), (2, 1, 1, 5, 0), (8, 8, 70001, 3, 9)])
def test_padded_copy_with_moments(gpu, rows, group, n, zl, zr):
    """mia_pad_rows_moments: the zero-padded copy bit for bit, the sums of each group of rows against
    double-precision sums; mia_normalize_rows_sums with them equals mia_normalize_rows."""
    rng = np.random.default_rng(rows * 10 + group)
    x = _dev((rng.standard_normal((rows, n)) * (1 + np.arange(rows))[:, None] + 0.3).astype(np.float32), gpu)
    lib = _lib.lib()
    s = torch.cuda.current_stream(gpu).cuda_stream
    out = torch.full((rows, zl + n + zr), 7.0, device=gpu)
    sums = torch.full((rows // group, 2), 5.0, dtype=torch.float64, device=gpu)
    _lib.check(lib.mia_pad_rows_moments(x.data_ptr(), rows, n, zl, zr, group, out.data_ptr(), sums.data_ptr(), s))
    assert torch.equal(out, torch.nn.functional.pad(x, (zl, zr)))
    xd = x.to(torch.float64).reshape(rows // group, -1)
    np.testing.assert_allclose(sums[:, 0].cpu().numpy(), xd.sum(dim=1).cpu().numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(sums[:, 1].cpu().numpy(), (xd * xd).sum(dim=1).cpu().numpy(), rtol=1e-11)
    g, m = rows // group, group * n
    a, b = torch.empty((g, m), device=gpu), torch.empty((g, m), device=gpu)
    sa, sb = torch.empty((g, 2), device=gpu), torch.empty((g, 2), device=gpu)
    scratch = torch.empty((g, 2), dtype=torch.float64, device=gpu)
    _lib.check(lib.mia_normalize_rows(x.data_ptr(), g, m, 1e-5, a.data_ptr(), sa.data_ptr(), scratch.data_ptr(), s))
    _lib.check(lib.mia_normalize_rows_sums(x.data_ptr(), g, m, 1e-5, b.data_ptr(), sb.data_ptr(), sums.data_ptr(), s))
    np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sb.cpu().numpy(), sa.cpu().numpy(), rtol=1e-6)
    assert lib.mia_pad_rows_moments(x.data_ptr(), rows, n, zl, zr, rows + 1, out.data_ptr(), sums.data_ptr(), s) \
        == _lib.MIA_ERR_INVALID

