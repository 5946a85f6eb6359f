"""Spectral gating on the GPU against the oracle (NumPy + scipy.ndimage, the reference's own
host arithmetic for the mask), through the C ABI.  The mask steps are compared one by one on
identical inputs (1e-6: float32 sums in another order), the whole functions on waveforms
(2e-4 of the peak: the STFT pair's rounding enters the choice of the quietest frames only through
values, the frame sets are compared explicitly)."""

import numpy as np
import pytest
import torch
from scipy.ndimage import minimum_filter1d, uniform_filter

import oracle.spectral_gate as orc
from mlx_audio_amd import _lib
from mlx_audio_amd.primitives import spectral_gate, spectral_gate_adaptive
from mlx_audio_amd.primitives.spectral_gate import _Ops

pytestmark = pytest.mark.gpu


def _dev(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def _noisy(rows=2, n=22050, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 22050.0
    y = np.stack([np.sin(2 * np.pi * (330 + 110 * r) * t) * (t > 0.25 + 0.1 * r) for r in range(rows)])
    return (y + 0.05 * rng.standard_normal((rows, n))).astype(np.float32)


def test_mask_building_blocks(gpu):
    rng = np.random.default_rng(1)
    mag = np.abs(rng.standard_normal((3, 65, 120))).astype(np.float32)
    ops = _Ops(gpu)
    d = _dev(mag, gpu)
    np.testing.assert_allclose(ops.frame_energy(d).cpu().numpy(), np.mean(mag ** 2, axis=1), rtol=2e-6)
    want = np.mean(mag, axis=-1) + 1.5 * np.std(mag, axis=-1)
    np.testing.assert_allclose(ops.noise_threshold(d, None, 1.5).cpu().numpy(), want, rtol=3e-6)
    idx = np.stack([rng.permutation(120)[:12] for _ in range(3)]).astype(np.int32)
    sel = np.stack([mag[b][:, idx[b]] for b in range(3)])
    want = np.mean(sel, axis=-1) + 2.0 * np.std(sel, axis=-1)
    np.testing.assert_allclose(ops.noise_threshold(d, _dev(idx, gpu), 2.0).cpu().numpy(), want, rtol=3e-6)
    for size in (1, 2, 5, 18, 300):
        np.testing.assert_array_equal(ops.time_min(d, size).cpu().numpy(),
                                      minimum_filter1d(mag, size=size, axis=-1, mode="nearest"))
    for fb, tf in ((9, 1), (2, 4), (3, 5), (1, 1), (70, 130)):
        np.testing.assert_allclose(ops.box(d, fb, tf).cpu().numpy(),
                                   uniform_filter(mag, size=(1, fb, tf), mode="nearest"), rtol=0, atol=1e-6)
    thr = np.abs(rng.standard_normal((3, 65))).astype(np.float32)
    want = np.clip((mag - thr[:, :, None] * 0.1) / (mag + 1e-8), 0.0, 1.0)
    np.testing.assert_allclose(ops.mask(d, _dev(thr, gpu), 0.1).cpu().numpy(), want, rtol=0, atol=1e-6)
    want1 = np.clip((mag - thr[:1, :, None] * 0.1) / (mag + 1e-8), 0.0, 1.0)
    np.testing.assert_allclose(ops.mask(d, _dev(thr[:1], gpu), 0.1).cpu().numpy(), want1, rtol=0, atol=1e-6)
    full = np.abs(rng.standard_normal(mag.shape)).astype(np.float32)
    want = np.clip((mag - full * 0.5) / (mag + 1e-8), 0.0, 1.0)
    np.testing.assert_allclose(ops.mask(d, _dev(full, gpu), 0.5).cpu().numpy(), want, rtol=0, atol=1e-6)
    m = ops.mask(d, _dev(full, gpu), 0.5)
    np.testing.assert_allclose(ops.keep(m, 0.8).cpu().numpy(), 1.0 - 0.8 * (1.0 - want), rtol=0, atol=1e-6)
    np.testing.assert_array_equal(ops.times(d, m).cpu().numpy(), mag * m.cpu().numpy())
    ph = rng.uniform(-np.pi, np.pi, mag.shape).astype(np.float32)
    mk = m.cpu().numpy()
    # the fused forms are bit-identical to the separate launches
    for keep, pd, size in ((False, 1.0, 1), (True, 0.7, 1), (False, 1.0, 9), (True, 0.7, 3), (False, 1.0, 200)):
        sep = ops.mask(d, _dev(thr, gpu), 0.1)
        if keep:
            sep = ops.keep(sep, pd)
        sep = ops.box(sep, size, 1)
        assert torch.equal(ops.mask_box(d, _dev(thr, gpu), 0.1, keep, pd, size), sep)
    assert torch.equal(ops.mask_box(d, _dev(full, gpu), 0.5, False, 1.0, 4), ops.box(ops.mask(d, _dev(full, gpu), 0.5), 4, 1))
    assert torch.equal(ops.mask_box(d, _dev(thr[:1], gpu), 0.1, False, 1.0, 2), ops.box(ops.mask(d, _dev(thr[:1], gpu), 0.1), 2, 1))
    php = _dev(rng.uniform(-np.pi, np.pi, mag.shape).astype(np.float32), gpu)
    for size in (2, 4, 5, 150):
        assert torch.equal(ops.apply(d, m, php, True, 0.9, size), ops.apply(d, ops.box(m, 1, size), php, True, 0.9))
    for keep, pd in ((True, 0.8), (False, 1.0)):
        got = ops.apply(d, m, _dev(ph, gpu), keep, pd)
        assert got.shape == mag.shape and got.transpose(1, 2).is_contiguous()
        mm = 1.0 - pd * (1.0 - mk) if keep else mk
        want = (mag * mm * np.exp(1j * ph.astype(np.float64))).astype(np.complex64)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=2e-6)

    # the spectrum form: mag * exp(i ph) handed over as the complex spectrum itself
    spec = (mag * np.exp(1j * ph.astype(np.float64))).astype(np.complex64)
    spec_d = _dev(spec.transpose(0, 2, 1), gpu).transpose(1, 2)   # a view of (rows, frames, bins) rows
    for keep, pd, size in ((True, 0.8, 1), (False, 1.0, 1), (True, 0.9, 4), (False, 1.0, 150)):
        got = ops.apply_spectrum(spec_d, m, keep, pd, size)
        assert got.shape == mag.shape and got.transpose(1, 2).is_contiguous()
        np.testing.assert_allclose(got.cpu().numpy(), ops.apply(d, m, _dev(ph, gpu), keep, pd, size).cpu().numpy(),
                                   rtol=0, atol=2e-6)
        mb = ops.box(m, 1, size).cpu().numpy()
        mm = 1.0 - pd * (1.0 - mb) if keep else mb
        if keep:   # the keep step contracts to one fused multiply-add on the device
            np.testing.assert_allclose(got.cpu().numpy(), spec * mm.astype(np.float32), rtol=1e-6, atol=1e-7)
        else:
            np.testing.assert_array_equal(got.cpu().numpy(), spec * mm.astype(np.float32))
    np.testing.assert_array_equal(ops.apply_spectrum(_dev(spec, gpu), m, False, 1.0).cpu().numpy(), spec * mk)
    for sf, st in ((2, 4), (9, 1), (3, 5), (70, 2)):   # a frequency average folded in: bit-identical
        assert torch.equal(ops.apply_spectrum(spec_d, m, True, 0.7, st, sf),
                           ops.apply_spectrum(spec_d, ops.box(m, sf, 1), True, 0.7, st))


@pytest.mark.parametrize("shape", [(1, 3, 2), (2, 1, 1), (1, 33, 31), (2, 32, 64), (1, 70, 5)])
def test_tiled_kernels_on_small_and_ragged_shapes(gpu, shape):
    """Windows wider than the array, single bins / frames, tiles cut by both edges: the LDS-tiled
    mask and closing kernels against SciPy's filters and against the one-thread-per-output forms
    (window 40 takes those)."""
    rng = np.random.default_rng(sum(shape))
    mag = np.abs(rng.standard_normal(shape)).astype(np.float32)
    thr = np.abs(rng.standard_normal(shape[:2])).astype(np.float32)
    spec = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    ops = _Ops(gpu)
    d, t = _dev(mag, gpu), _dev(thr, gpu)
    spec_d = _dev(spec.transpose(0, 2, 1), gpu).transpose(1, 2)
    base = np.clip((mag - thr[:, :, None] * 0.3) / (mag + 1e-8), 0.0, 1.0)
    for size in (1, 2, 9, 33, 40):
        got = ops.mask_box(d, t, 0.3, False, 1.0, size).cpu().numpy()
        np.testing.assert_allclose(got, uniform_filter(base, size=(1, size, 1), mode="nearest"), rtol=0, atol=1e-6)
        assert torch.equal(ops.mask_box(d, t, 0.3, True, 0.6, size), ops.box(ops.keep(ops.mask(d, t, 0.3), 0.6), size, 1))
    m = ops.mask(d, t, 0.3)
    for sf, st in ((1, 1), (2, 4), (9, 33), (33, 2), (1, 40), (40, 3)):
        got = ops.apply_spectrum(spec_d, m, False, 1.0, st, sf)
        want = spec * uniform_filter(uniform_filter(base, size=(1, sf, 1), mode="nearest"),
                                     size=(1, 1, st), mode="nearest")
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=3e-6 * max(1.0, np.abs(spec).max()))
        assert torch.equal(got, ops.apply_spectrum(spec_d, ops.box(ops.box(m, sf, 1), 1, st), False, 1.0))


@pytest.mark.parametrize("shape", [(3, 65, 120), (1, 3, 2), (2, 1025, 44), (2, 33, 64), (1, 40, 1)])
def test_steps_taken_from_the_spectrum(gpu, shape):
    """Frame energy, noise threshold and mask + frequency average from the complex (rows, frames,
    bins) spectrum against the same steps on its magnitude plane: threshold and mask bit for bit,
    the energy (another summation order) to float32 rounding."""
    from mlx_audio_amd.primitives import magnitude

    rows, bins, frames = shape
    rng = np.random.default_rng(bins * 7 + frames)
    spec = (rng.standard_normal((rows, frames, bins)) + 1j * rng.standard_normal((rows, frames, bins))).astype(np.complex64)
    spec_rows = _dev(spec, gpu)
    mag = magnitude(spec_rows.transpose(1, 2))            # (rows, bins, frames) plane
    ops = _Ops(gpu)
    np.testing.assert_allclose(ops.frame_energy_spec(spec_rows).cpu().numpy(), ops.frame_energy(mag).cpu().numpy(),
                               rtol=3e-6)
    n_sel = max(1, frames // 3)
    idx = _dev(np.stack([rng.permutation(frames)[:n_sel] for _ in range(rows)]).astype(np.int32), gpu)
    for sel in (idx, None):
        assert torch.equal(ops.noise_threshold_spec(spec_rows, sel, 1.5), ops.noise_threshold(mag, sel, 1.5))
    for thr_rows in (rows, 1):
        thr = _dev(np.abs(rng.standard_normal((thr_rows, bins))).astype(np.float32), gpu)
        for size, keep in ((1, False), (2, True), (9, False), (33, True)):
            assert torch.equal(ops.mask_box_spec(spec_rows, thr, 0.4, keep, 0.7, size),
                               ops.mask_box(mag, thr, 0.4, keep, 0.7, size)), (thr_rows, size, keep)
    lib = _lib.lib()
    s = torch.cuda.current_stream(gpu).cuda_stream
    out = torch.empty((rows, bins, frames), device=gpu)
    thr = torch.ones((rows, bins), device=gpu)
    assert lib.mia_gate_mask_box_spec(spec_rows.data_ptr(), thr.data_ptr(), rows, bins, frames, rows, 1.0, 0, 1.0,
                                      34, out.data_ptr(), s) == _lib.MIA_ERR_UNSUPPORTED
    assert lib.mia_gate_mask_box_spec(spec_rows.data_ptr(), thr.data_ptr(), rows, bins, frames, rows + 1, 1.0, 0, 1.0,
                                      3, out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_frame_energy_spec(0, rows, bins, frames, out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_noise_threshold_spec(spec_rows.data_ptr(), rows, bins, frames, idx.data_ptr(), 0, 1.0,
                                        out.data_ptr(), s) == _lib.MIA_ERR_INVALID


def test_whole_mask_matches_the_oracle(gpu):
    """gate_mask of the oracle against the same sequence of launches on identical magnitudes."""
    rng = np.random.default_rng(2)
    mag = np.abs(rng.standard_normal((2, 1025, 44))).astype(np.float32)
    ops = _Ops(gpu)
    d = _dev(mag, gpu)
    # non-stationary branch
    want = orc.gate_mask(mag, stationary=False)
    tc = max(1, int(0.05 * 22050 / 512))
    m = ops.mask(d, ops.time_min(d, tc), 10 ** (-20 / 20.0) * 2.5)
    m = ops.keep(ops.box(ops.box(m, 9, 1), 2, 4), 1.0)
    np.testing.assert_allclose(m.cpu().numpy(), want, rtol=0, atol=2e-6)
    want = orc.adaptive_mask(mag)
    a = max(1, int(100.0 * 22050 / (1000 * 512)))
    m = ops.box(ops.keep(ops.mask(d, ops.time_min(d, 2 * a + 1), 1.1), 1.0), 3, 5)
    np.testing.assert_allclose(m.cpu().numpy(), want, rtol=0, atol=2e-6)


@pytest.mark.parametrize("kw", [{}, {"stationary": False}, {"prop_decrease": 0.6, "threshold_db": -10.0},
                                {"n_fft": 1024, "hop_length": 256, "freq_smooth_hz": 0.0, "time_smooth_ms": 0.0},
                                {"n_grad_freq": 1, "n_grad_time": 1, "time_smooth_ms": 60.0},
                                {"window": "hamming", "win_length": 1500}])
def test_spectral_gate_matches_the_oracle(gpu, kw):
    y = _noisy()
    got = spectral_gate(_dev(y, gpu), **kw).cpu().numpy()
    want = orc.spectral_gate(y, **kw)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()
    one = spectral_gate(_dev(y[0], gpu), **kw).cpu().numpy()
    assert np.abs(one - orc.spectral_gate(y[0], **kw)).max() <= 2e-4 * np.abs(want).max()


@pytest.mark.parametrize("kw", [{}, {"freq_smooth_hz": 4000.0}])   # direct (spectrum rows) / magnitude plane
def test_quiet_frame_choice_with_ties(gpu, kw):
    """Runs of digital silence and repeated stretches give frames with exactly equal energy, so
    which of them count as "the quietest tenth" depends on how ties are broken (the reference:
    NumPy's argsort on the host; here a stable device sort, on energies summed in another
    order).  Tied frames of this kind are equal frames, so the noise statistics - and with them
    the output - must not depend on the choice: both branches against the oracle."""
    rng = np.random.default_rng(5)
    n = 22050 * 2
    y = np.zeros((2, n), np.float32)
    burst = (0.3 * rng.standard_normal(4096)).astype(np.float32)
    for start in (6000, 14192, 22384):            # the same burst three times: tied loud frames
        y[0, start:start + 4096] = burst
    y[0, 32000:] = (0.01 * rng.standard_normal(n - 32000)).astype(np.float32)
    y[1, :] = np.tile((0.05 * rng.standard_normal(2048)).astype(np.float32), n // 2048 + 1)[:n]  # periodic: every frame ties
    y[1, 20000:24000] = 0.0
    got = spectral_gate(_dev(y, gpu), **kw).cpu().numpy()
    want = orc.spectral_gate(y, **kw)
    assert np.abs(got - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-3)


def test_noise_profile_and_adaptive(gpu):
    y = _noisy()
    for prof in (y[0, :4000], y[:, :4000]):
        got = spectral_gate(_dev(y, gpu), noise_profile=_dev(prof, gpu)).cpu().numpy()
        want = orc.spectral_gate(y, noise_profile=prof)
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()
    for kw in ({}, {"threshold_db": -6.0, "prop_decrease": 0.7, "look_ahead_ms": 30.0}, {"n_fft": 512}):
        got = spectral_gate_adaptive(_dev(y, gpu), **kw).cpu().numpy()
        want = orc.spectral_gate_adaptive(y, **kw)
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()
    with pytest.raises(ValueError, match="noise_profile has 3 rows"):
        spectral_gate(_dev(y, gpu), noise_profile=_dev(np.zeros((3, 4000), np.float32), gpu))


def test_behaviour_and_c_abi_checks(gpu):
    y = _noisy(rows=1)[0]
    out = spectral_gate(_dev(y, gpu)).cpu().numpy()
    assert np.abs(out[:4000]).mean() < 0.9 * np.abs(y[:4000]).mean()
    np.testing.assert_allclose(spectral_gate(_dev(y, gpu), prop_decrease=0.0).cpu().numpy(), y, atol=1e-5)
    lib = _lib.lib()
    a = torch.ones((2, 4, 6), device=gpu)
    o = torch.zeros((2, 4, 6), device=gpu)
    s = torch.cuda.current_stream(gpu).cuda_stream
    assert lib.mia_time_min_filter(a.data_ptr(), 8, 6, 0, o.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_time_min_filter(a.data_ptr(), 8, 6, 3, a.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_box_filter(a.data_ptr(), 2, 4, 6, 0, o.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_gate_mask(a.data_ptr(), a.data_ptr(), 2, 4, 6, 0, 1, 1.0, o.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_mask_op(a.data_ptr(), 0, 48, 1, 0.0, o.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_mask_op(a.data_ptr(), 0, 48, 2, 0.0, o.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_noise_threshold(a.data_ptr(), 2, 4, 0, 0, 0, 1.0, o.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_frame_energy(a.data_ptr(), 0, 4, 6, o.data_ptr(), s) == _lib.MIA_OK
    assert lib.mia_gate_apply_spectrum(a.data_ptr(), a.data_ptr(), 1, 4, 3, 0, 1.0, 0, 1, o.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_gate_apply_spectrum(a.data_ptr(), a.data_ptr(), 1, 4, 3, 0, 1.0, 1, 0, o.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_gate_apply_spectrum(0, a.data_ptr(), 1, 4, 3, 0, 1.0, 1, 1, o.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_gate_apply_spectrum(0, 0, 0, 4, 3, 0, 1.0, 1, 1, 0, s) == _lib.MIA_OK


_ONE_LAUNCH_CASES = [
    {}, {"n_fft": 1024}, {"n_fft": 4096, "sr": 44100}, {"sr": 16000}, {"sr": 44100, "n_grad_freq": 3, "n_grad_time": 5, "prop_decrease": 0.8},
    {"n_grad_freq": 1, "n_grad_time": 1}, {"freq_smooth_hz": 0.0, "time_smooth_ms": 0.0},
    {"n_grad_freq": 4, "n_grad_time": 8, "freq_smooth_hz": 172.0}, {"n_fft": 4096, "sr": 44100, "n_grad_time": 6},
    {"noise": True},
    {"window": "hamming", "win_length": 1500, "prop_decrease": 0.5},
]


@pytest.mark.parametrize("shape", [(1, 44100), (3, 30000), (2, 1500), (1, 512), (7, 100000)])
def test_one_launch_gate_and_inverse_equals_the_separate_launches(gpu, shape):
    """The stationary gate whose mask pipeline runs inside the inverse STFT kernel (neither mask
    nor gated spectrum in memory) against the same call on the separate kernels: rows shorter
    than a tile, rows that end inside a tile, several rows per workgroup, every window shape the
    kernel takes; equal to float32 rounding of the mask values."""
    import importlib
    mod = importlib.import_module("mlx_audio_amd.primitives.spectral_gate")
    g = torch.Generator(device=gpu)
    g.manual_seed(shape[0] * 1000 + shape[1])
    x = 0.05 * torch.randn(shape, generator=g, device=gpu)
    t = torch.arange(shape[1], device=gpu) / 22050.0
    x = x + 0.5 * torch.sin(2 * np.pi * 440 * t) * (t > 0.4 * shape[1] / 22050.0)
    limits = mod._FUSED_MAX_FRAMES, mod._FUSED_MAX_FRAMES_4096
    for case in _ONE_LAUNCH_CASES:
        kw = dict(case)
        if kw.pop("noise", False):
            kw["noise_profile"] = 0.05 * torch.randn(8000, generator=g, device=gpu)
        assert mod._FUSE_INVERSE
        try:
            mod._FUSED_MAX_FRAMES = mod._FUSED_MAX_FRAMES_4096 = 8   # every time window the kernel takes
            one = spectral_gate(x, **kw)
            mod._FUSE_INVERSE = False
            three = spectral_gate(x, **kw)
        finally:
            mod._FUSE_INVERSE = True
            mod._FUSED_MAX_FRAMES, mod._FUSED_MAX_FRAMES_4096 = limits
        assert one.shape == three.shape == x.shape
        assert (one - three).abs().max().item() <= 1e-6 * max(three.abs().max().item(), 1e-3), case


@pytest.mark.parametrize("n_fft,hop", [(2048, 512), (1024, 256), (4096, 1024), (2048, 300)])
def test_frame_energies_from_the_forward_kernel(gpu, n_fft, hop):
    """mia_stft_energy: the spectrum bit for bit mia_stft's, the energies those of the separate
    kernel on that spectrum to float32 rounding, rows independent of the batch around them, rows
    that end inside a tile and clips shorter than a frame included."""
    from mlx_audio_amd.primitives import stft
    g = torch.Generator(device=gpu)
    g.manual_seed(n_fft + hop)
    for rows, n in [(3, 30000), (1, 700), (5, n_fft * 3 + 17)]:
        x = torch.randn((rows, n), generator=g, device=gpu)
        ops = _Ops(torch.device(gpu))
        spec, energy = ops.stft_energy(x, n_fft, hop, n_fft, "hann")
        ref = stft(x, n_fft=n_fft, hop_length=hop)
        assert torch.equal(torch.view_as_real(spec.contiguous()), torch.view_as_real(ref.contiguous()))
        want = ops.frame_energy_spec(ref.transpose(1, 2))
        np.testing.assert_allclose(energy.cpu().numpy(), want.cpu().numpy(), rtol=3e-6)
        mag = np.abs(ref.cpu().numpy())
        np.testing.assert_allclose(energy.cpu().numpy(), np.mean(mag.astype(np.float64) ** 2, axis=1), rtol=3e-6)
        alone = ops.stft_energy(x[rows - 1:].contiguous(), n_fft, hop, n_fft, "hann")[1]
        assert torch.equal(alone[0], energy[rows - 1])
    lib = _lib.lib()
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives.windows import padded_window_np
    small = get_plan(torch.device(gpu), 512, 128, True, "constant", padded_window_np("hann", 512, 512))
    assert lib.mia_stft_energy_supported(small.handle) == 0
    assert _Ops(torch.device(gpu)).stft_energy(torch.zeros((1, 4000), device=gpu), 512, 128, 512, "hann") is None
    s = torch.cuda.current_stream(gpu).cuda_stream
    z = torch.zeros((1, 4000), device=gpu)
    assert lib.mia_stft_energy(small.handle, z.data_ptr(), 1, 4000, 1.0, z.data_ptr(), z.data_ptr(), s) == _lib.MIA_ERR_UNSUPPORTED


def test_quietest_frames_are_a_stable_sorts_first(gpu):
    """mia_quietest_frames against torch's stable sort: random rows, rows full of ties, NaN and
    infinities, one frame, n_selected = frames; long rows fall back to the sort."""
    ops = _Ops(torch.device(gpu))
    g = torch.Generator(device=gpu)
    g.manual_seed(11)
    for rows, frames in [(3, 431), (1, 1), (2, 7), (5, 1000), (1, 8192)]:
        e = torch.rand((rows, frames), generator=g, device=gpu)
        e[:, ::3] = torch.round(e[:, ::3] * 4) / 4          # many exact ties
        if frames > 6:
            e[0, 2] = float("nan"); e[0, 5] = float("inf"); e[-1, 1] = 0.0; e[-1, 4] = -0.0
        for n_sel in sorted({1, max(1, frames // 10), frames}):
            got = ops.quietest(e, n_sel)
            want = torch.sort(e, dim=1, stable=True).indices[:, :n_sel].to(torch.int32)
            assert got.dtype == torch.int32 and torch.equal(got, want), (rows, frames, n_sel)
    long_row = torch.rand((1, 9000), generator=g, device=gpu)
    assert torch.equal(ops.quietest(long_row, 900),
                       torch.sort(long_row, dim=1, stable=True).indices[:, :900].to(torch.int32))
    lib = _lib.lib()
    s = torch.cuda.current_stream(gpu).cuda_stream
    out = torch.zeros((1, 9000), dtype=torch.int32, device=gpu)
    assert lib.mia_quietest_frames(long_row.data_ptr(), 1, 9000, 10, out.data_ptr(), s) == _lib.MIA_ERR_UNSUPPORTED
    assert lib.mia_quietest_frames(long_row.data_ptr(), 1, 100, 0, out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_quietest_frames(long_row.data_ptr(), 1, 100, 101, out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_quietest_frames(0, 0, 100, 10, 0, s) == _lib.MIA_OK


def test_one_launch_gate_is_the_path_taken(gpu):
    """The default call's framing and windows have the one-launch kernel, wider windows and other
    sizes say so and fall back (still right against the oracle), and the entry point checks its
    arguments."""
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives.windows import padded_window_np
    lib = _lib.lib()
    win = padded_window_np("hann", 2048, 2048)
    plan = get_plan(torch.device(gpu), 2048, 512, True, "constant", win)
    assert lib.mia_istft_gated_supported(plan.handle, 9, 2, 4) == 1
    assert lib.mia_istft_gated_supported(plan.handle, 16, 4, 8) == 1
    assert lib.mia_istft_gated_supported(plan.handle, 17, 2, 4) == 0
    assert lib.mia_istft_gated_supported(plan.handle, 9, 5, 4) == 0
    assert lib.mia_istft_gated_supported(plan.handle, 9, 2, 9) == 0
    assert lib.mia_istft_gated_supported(plan.handle, 0, 2, 4) == 0
    other = get_plan(torch.device(gpu), 2048, 256, True, "constant", win)
    assert lib.mia_istft_gated_supported(other.handle, 9, 2, 4) == 0
    small = get_plan(torch.device(gpu), 512, 128, True, "constant", padded_window_np("hann", 512, 512))
    assert lib.mia_istft_gated_supported(small.handle, 3, 2, 4) == 0
    assert _Ops.fused_windows([("f", 9), ("f", 2), ("t", 4)]) == (9, 2, 4)
    assert _Ops.fused_windows([("f", 9), ("t", 2), ("f", 2), ("t", 4)]) is None
    assert _Ops.fused_windows([("t", 4)]) == (1, 1, 4)
    assert _Ops.fused_windows([]) == (1, 1, 1)
    # time windows the kernel takes but runs slower than the separate launches: the mirror goes the long way
    ops = _Ops(torch.device(gpu))
    C8 = torch.zeros((1, 6, 1025), dtype=torch.complex64, device=gpu).transpose(1, 2)
    thr8 = torch.ones((1, 1025), device=gpu)
    assert ops.gated_istft(C8, thr8, 0.1, (9, 2, 8), 1.0, 2048, 512, 2048, "hann", 2000) is None
    assert ops.gated_istft(C8, thr8, 0.1, (9, 2, 7), 1.0, 2048, 512, 2048, "hann", 2000) is not None
    s = torch.cuda.current_stream(gpu).cuda_stream
    spec = torch.zeros((2, 5, 1025), dtype=torch.complex64, device=gpu)
    thr = torch.ones((2, 1025), device=gpu)
    out = torch.full((2, 4096), 7.0, device=gpu)
    args = (1.0, thr.data_ptr(), 2, 0.1, 1, 1.0)
    assert lib.mia_istft_gated(plan.handle, spec.data_ptr(), 2, 5, 4096, *args, 9, 2, 4, out.data_ptr(), s) == _lib.MIA_OK
    assert torch.count_nonzero(out).item() == 0
    assert lib.mia_istft_gated(plan.handle, spec.data_ptr(), 2, 5, 4096, *args, 17, 2, 4, out.data_ptr(), s) == _lib.MIA_ERR_UNSUPPORTED
    assert lib.mia_istft_gated(plan.handle, spec.data_ptr(), 2, 5, 4096, 1.0, thr.data_ptr(), 3, 0.1, 1, 1.0, 9, 2, 4,
                               out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    assert lib.mia_istft_gated(plan.handle, 0, 2, 5, 4096, *args, 9, 2, 4, out.data_ptr(), s) == _lib.MIA_ERR_INVALID
    out.fill_(7.0)
    assert lib.mia_istft_gated(plan.handle, spec.data_ptr(), 2, 0, 4096, *args, 9, 2, 4, out.data_ptr(), s) == _lib.MIA_OK
    assert torch.count_nonzero(out).item() == 0   # no frames: silence
    # a window wider than the kernel's goes the long way and still matches the oracle
    y = _noisy()
    kw = {"freq_smooth_hz": 250.0}
    got = spectral_gate(_dev(y, gpu), **kw).cpu().numpy()
    want = orc.spectral_gate(y, **kw)
    assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()


def test_full_size_batch(gpu):
    """64 x 10 s @22.05 kHz: spot rows against the oracle, batch invariance bit for bit."""
    g = torch.Generator(device=gpu)
    g.manual_seed(2)
    y = 0.05 * torch.randn((64, 220500), generator=g, device=gpu)
    t = torch.arange(220500, device=gpu) / 22050.0
    y = y + torch.sin(2 * np.pi * 440 * t) * (t > 2.0)
    out = spectral_gate(y)
    assert out.shape == y.shape
    rows = [0, 63]
    want = orc.spectral_gate(y[rows].cpu().numpy())
    assert np.abs(out[rows].cpu().numpy() - want).max() <= 2e-4 * np.abs(want).max()
    assert torch.equal(spectral_gate(y[5:9]), out[5:9])
    assert out[:, :22050].abs().mean() < 0.9 * y[:, :22050].abs().mean()
