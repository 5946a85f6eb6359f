"""The n_fft = 400 kernel's forms against each other: the compiled mel sweep (Whisper's 80- and
128-filter banks as straight-line code) gives the same bits as the table-driven sweep on the
same transform; the LDS-DMA kernel (prime-factor transform, line-aligned stores) agrees with
the register-staging kernel to float32 round-off of the transform."""

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from mlx_audio_amd import _lib  # noqa: E402

pytestmark = pytest.mark.gpu


def _plan(n_mels):
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives.mel import mel_filterbank_np
    from mlx_audio_amd.primitives.windows import padded_window_np

    basis = mel_filterbank_np(16000, 400, n_mels, 0.0, 8000.0, False, "slaney")
    return get_plan(torch.device("cuda", 0), 400, 160, True, "constant",
                    padded_window_np("hann", 400, 400), mel_basis=basis, power=2.0)


def _logmel(plan, y, per_clip=1):
    lib = _lib.lib()
    B, T = y.shape
    out = torch.empty((B, plan.n_mels, plan.num_frames(T)), dtype=torch.float32, device=y.device)
    ws_bytes = lib.mia_mel_workspace_bytes(plan.handle, B, T)
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=y.device)
    _lib.check(lib.mia_melspectrogram(plan.handle, y.data_ptr(), B, T, _lib.EPI_WHISPER, 1.0, 1e-10,
                                      10.0, -1.0, per_clip, out.data_ptr(), ws.data_ptr(), ws_bytes,
                                      torch.cuda.current_stream(y.device).cuda_stream))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("n_mels", [80, 128])
def test_whisper_banks_take_the_compiled_sweep(n_mels):
    plan = _plan(n_mels)
    lib = _lib.lib()
    assert lib.mia_plan_get_info(plan.handle, b"w400_tables") == 1
    assert lib.mia_plan_get_info(plan.handle, b"w400_bank") == n_mels
    assert lib.mia_plan_get_info(plan.handle, b"compute_units") > 0


def test_other_banks_do_not():
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives.mel import mel_filterbank_np
    from mlx_audio_amd.primitives.windows import padded_window_np

    basis = mel_filterbank_np(16000, 400, 64, 20.0, 7600.0, False, "slaney")
    plan = get_plan(torch.device("cuda", 0), 400, 160, True, "constant",
                    padded_window_np("hann", 400, 400), mel_basis=basis, power=2.0)
    assert _lib.lib().mia_plan_get_info(plan.handle, b"w400_bank") == 0


@pytest.mark.parametrize("n_mels", [80, 128])
@pytest.mark.parametrize("shape", [(3, 480000), (5, 16000), (2, 10239), (1, 160 * 64 * 3 - 200),
                                   (300, 4000)])
def test_compiled_sweep_equals_table_sweep(n_mels, shape):
    plan = _plan(n_mels)
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(shape[0] * 1000 + n_mels)
    y = torch.randn(shape, device="cuda", generator=g)
    y[0, : shape[1] // 2] *= 1e-4  # a clip whose floor is active
    if shape[0] > 1:
        y[1] = 0.0  # silence
    try:
        for per_clip in (1, 0):
            got = _logmel(plan, y, per_clip)  # LDS-DMA kernel where the rows are 16-byte aligned
            _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_vgpr_staging", 1))
            compiled = _logmel(plan, y, per_clip)  # register staging, compiled sweep
            _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_table_sweep", 1))
            table = _logmel(plan, y, per_clip)  # register staging, table-driven sweep
            _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_table_sweep", 0))
            _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_vgpr_staging", 0))
            assert np.array_equal(compiled, table), (n_mels, shape, per_clip)
            # two float32 transforms of the same frames: (log10 + 4) / 4 agrees to ~1e-6 where
            # the mel power is not a cancellation residue
            assert np.abs(got - table).max() <= 2e-5, (n_mels, shape, per_clip)
    finally:
        lib.mia_plan_set_option(plan.handle, b"w400_table_sweep", 0)
        lib.mia_plan_set_option(plan.handle, b"w400_vgpr_staging", 0)


def test_nan_and_inf_clips_compiled_sweep():
    plan = _plan(128)
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    y = torch.randn((4, 32000), device="cuda", generator=g)
    y[1, 777] = float("nan")
    y[2, 1234] = float("inf")
    got = _logmel(plan, y)
    _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_table_sweep", 1))
    try:
        want = _logmel(plan, y)
    finally:
        lib.mia_plan_set_option(plan.handle, b"w400_table_sweep", 0)
    assert np.allclose(got, want, rtol=0, atol=2e-5, equal_nan=True)
    assert np.isnan(got[1]).all() and np.isfinite(got[0]).all() and np.isfinite(got[3]).all()


@pytest.mark.parametrize("n_mels", [80, 128, 64])
@pytest.mark.parametrize("shape", [(3, 480000), (5, 16000), (2, 10239), (300, 4000)])
def test_matrix_core_projection_equals_sweep(n_mels, shape):
    """The A/B variant (v_mfma_f32_16x16x4_f32 over each 16-filter band) against the sweep on
    the same transform: the same products, summed in bin order by both; only the grouping of
    the four products inside one matrix instruction may differ."""
    if n_mels == 64:
        from mlx_audio_amd._runtime import get_plan
        from mlx_audio_amd.primitives.mel import mel_filterbank_np
        from mlx_audio_amd.primitives.windows import padded_window_np

        basis = mel_filterbank_np(16000, 400, 64, 20.0, 7600.0, False, "slaney")
        plan = get_plan(torch.device("cuda", 0), 400, 160, True, "constant",
                        padded_window_np("hann", 400, 400), mel_basis=basis, power=2.0)
    else:
        plan = _plan(n_mels)
    lib = _lib.lib()
    assert lib.mia_plan_get_info(plan.handle, b"w400_mfma_tables") == 1
    g = torch.Generator(device="cuda").manual_seed(shape[0] * 77 + n_mels)
    y = torch.randn(shape, device="cuda", generator=g)
    y[0, : shape[1] // 2] *= 1e-4
    y[1] = 0.0
    try:
        for per_clip in (1, 0):
            _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_vgpr_staging", 1))
            _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_table_sweep", 1))
            table = _logmel(plan, y, per_clip)
            _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_mfma_mel", 1))
            mfma = _logmel(plan, y, per_clip)
            _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_mfma_mel", 0))
            assert np.abs(mfma - table).max() <= 2e-6, (n_mels, shape, per_clip)
    finally:
        lib.mia_plan_set_option(plan.handle, b"w400_mfma_mel", 0)
        lib.mia_plan_set_option(plan.handle, b"w400_table_sweep", 0)
        lib.mia_plan_set_option(plan.handle, b"w400_vgpr_staging", 0)


def test_nan_and_inf_clips_matrix_core_projection():
    plan = _plan(128)
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(6)
    y = torch.randn((4, 32000), device="cuda", generator=g)
    y[1, 777] = float("nan")
    y[2, 1234] = float("inf")
    _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_table_sweep", 1))
    try:
        want = _logmel(plan, y)
        _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_mfma_mel", 1))
        got = _logmel(plan, y)
    finally:
        lib.mia_plan_set_option(plan.handle, b"w400_mfma_mel", 0)
        lib.mia_plan_set_option(plan.handle, b"w400_table_sweep", 0)
    assert np.allclose(got, want, rtol=0, atol=2e-6, equal_nan=True)
    assert np.isnan(got[1]).all() and np.isnan(got[2]).all() and np.isfinite(got[0]).all()


@pytest.mark.parametrize("n_mels", [80, 128])
@pytest.mark.parametrize("shape", [(3, 480000), (5, 16000), (2, 10240), (300, 4000)])
def test_cross_lane_pass2_equals_lds_exchange(n_mels, shape):
    """The A/B variant of the LDS-DMA kernel whose radix-16 pass runs across the lanes with
    DPP operands, against the shipped form (exchange through LDS, register butterflies): the
    same transform in another order of additions, so float32 round-off apart (each form is
    within 2e-6 of the float64 result, hence 4e-6 between the two)."""
    plan = _plan(n_mels)
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(shape[0] * 31 + n_mels)
    y = torch.randn(shape, device="cuda", generator=g)
    y[0, : shape[1] // 2] *= 1e-4
    y[1] = 0.0
    try:
        for per_clip in (1, 0):
            want = _logmel(plan, y, per_clip)
            _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_dpp_pass2", 1))
            got = _logmel(plan, y, per_clip)
            _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_dpp_pass2", 0))
            assert np.abs(got - want).max() <= 4e-6, (n_mels, shape, per_clip)
    finally:
        lib.mia_plan_set_option(plan.handle, b"w400_dpp_pass2", 0)


def test_nan_and_inf_clips_cross_lane_pass2():
    plan = _plan(128)
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(8)
    y = torch.randn((4, 32000), device="cuda", generator=g)
    y[1, 777] = float("nan")
    y[2, 1234] = float("inf")
    try:
        want = _logmel(plan, y)
        _lib.check(lib.mia_plan_set_option(plan.handle, b"w400_dpp_pass2", 1))
        got = _logmel(plan, y)
    finally:
        lib.mia_plan_set_option(plan.handle, b"w400_dpp_pass2", 0)
    assert np.allclose(got, want, rtol=0, atol=4e-6, equal_nan=True)
    assert np.isnan(got[1]).all() and np.isnan(got[2]).all() and np.isfinite(got[0]).all()


def test_paths_behind_the_headline_kernel_keep_parity():
    """What does NOT run on the LDS-DMA kernel at n_fft 400 / hop 160 - a filterbank other than
    Whisper's two, the dB epilogue - takes the register-staging kernel with the table-driven
    sweep; odd clip lengths (rows at every 4-byte phase) stay on the LDS-DMA kernel.  Each against
    the oracle, with the tolerance of the headline path (profiles/r03_w400_paths.txt has their times)."""
    import oracle.spectral as o
    from mlx_audio_amd.frontends import compute_log_mel_spectrogram
    from mlx_audio_amd.primitives import melspectrogram, power_to_db

    rng = np.random.default_rng(400)
    # odd length, batch: clip b starts b * 480001 floats in
    x = rng.standard_normal((6, 480001)).astype(np.float32)
    got = compute_log_mel_spectrogram(torch.from_numpy(x).cuda(), n_mels=128, per_clip=True).cpu().numpy()
    for b in (0, 3, 5):
        assert np.abs(got[b] - o.compute_log_mel_spectrogram(x[b], n_mels=128)).max() <= 1e-4, b
    # a 64-filter bank on the same framing, linear and dB
    kw = dict(sr=16000, n_fft=400, hop_length=160, n_mels=64, fmin=20.0, fmax=7600.0)
    y = x[:2, :160000]
    m = melspectrogram(torch.from_numpy(y).cuda(), **kw)
    want = o.melspectrogram(y, **kw)
    np.testing.assert_allclose(m.cpu().numpy(), want, rtol=1e-4, atol=1e-5 * want.max())
    db = power_to_db(m, top_db=80.0).cpu().numpy()
    assert np.abs(db - o.power_to_db(want, top_db=80.0)).max() <= 2e-3


@pytest.mark.parametrize("n_mels", [40, 128])
def test_magnitude_mel_on_the_n400_kernel(n_mels):
    """power = 1 at n_fft 400 / hop 160 (onset strength at Whisper's framing, reference
    onset.py:103-117): the n_fft = 400 kernel applies the filterbank to magnitudes.  Against the
    oracle and against the generic plan the same call takes at hop 161."""
    import oracle.spectral as o
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives import melspectrogram, onset_strength
    from mlx_audio_amd.primitives.mel import mel_filterbank_np
    from mlx_audio_amd.primitives.windows import padded_window_np
    import oracle.onset as oo

    rng = np.random.default_rng(n_mels)
    y = rng.standard_normal((3, 48000)).astype(np.float32)
    y[1] *= 1e-3
    kw = dict(sr=16000, n_fft=400, hop_length=160, n_mels=n_mels, power=1.0)
    got = melspectrogram(torch.from_numpy(y).cuda(), **kw).cpu().numpy()
    want = o.melspectrogram(y, **kw)
    for b in range(3):
        np.testing.assert_allclose(got[b], want[b], rtol=1e-4, atol=2e-6 * want[b].max())
    basis = mel_filterbank_np(16000, 400, n_mels, 0.0, 8000.0, False, "slaney")
    plan = get_plan(torch.device("cuda", 0), 400, 160, True, "constant", padded_window_np("hann", 400, 400),
                    mel_basis=basis, power=1.0)
    assert _lib.lib().mia_plan_get_info(plan.handle, b"w400_tables") == 1
    env = onset_strength(torch.from_numpy(y).cuda(), sr=16000, n_fft=400, hop_length=160, n_mels=n_mels).cpu().numpy()
    np.testing.assert_allclose(env, oo.onset_strength(y, sr=16000, n_fft=400, hop_length=160, n_mels=n_mels),
                               rtol=2e-3, atol=2e-5 * max(1.0, float(env.max())))
