"""mia_melspectrogram_static (the entry point that allocates nothing) against
mia_melspectrogram, through the C ABI: bit-identical outputs on two calls in a row,
the second with a smaller input, plus the size limit.  Collected last on purpose."""

import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from mlx_audio_amd import _lib  # noqa: E402

pytestmark = pytest.mark.gpu


def _whisper_plan(n_mels):
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives.mel import mel_filterbank_np
    from mlx_audio_amd.primitives.windows import padded_window_np

    basis = mel_filterbank_np(16000, 400, n_mels, 0.0, 8000.0, False, "slaney")
    win = padded_window_np("hann", 400, 400)
    return get_plan(torch.device("cuda", 0), 400, 160, True, "constant", win, mel_basis=basis,
                    power=2.0)


def _both(plan, y, epilogue, top_db, per_clip):
    """(workspace entry point, static entry point) outputs for one input."""
    lib = _lib.lib()
    B, T = y.shape
    F = plan.num_frames(T)
    stream = torch.cuda.current_stream(y.device).cuda_stream
    want = torch.empty((B, plan.n_mels, F), dtype=torch.float32, device=y.device)
    got = torch.full_like(want, float("nan"))
    ws_bytes = lib.mia_mel_workspace_bytes(plan.handle, B, T)
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=y.device)
    _lib.check(lib.mia_melspectrogram(plan.handle, y.data_ptr(), B, T, epilogue, 1.0, 1e-10, 10.0,
                                      top_db, per_clip, want.data_ptr(), ws.data_ptr(), ws_bytes,
                                      stream))
    _lib.check(lib.mia_melspectrogram_static(plan.handle, y.data_ptr(), B, T, epilogue, 1.0, 1e-10,
                                             10.0, top_db, per_clip, got.data_ptr(), stream))
    torch.cuda.synchronize()
    return want.cpu().numpy(), got.cpu().numpy()


@pytest.mark.parametrize("n_mels", [128, 80])
def test_static_entry_point_matches_workspace_entry_point(n_mels):
    plan = _whisper_plan(n_mels)
    g = torch.Generator(device="cuda").manual_seed(7)
    # tonal + weak noise: the clamp floor is active, so the keys matter
    t = torch.arange(48000, device="cuda", dtype=torch.float32) / 16000.0
    big = torch.sin(2 * np.pi * 440.0 * t)[None, :] + 1e-3 * torch.randn(
        (24, 48000), device="cuda", generator=g)
    big[3] = 0.0  # silence
    small = torch.randn((5, 16000), device="cuda", generator=g)  # fewer clips, shorter clips
    for y in (big.contiguous(), small.contiguous()):  # two calls in a row, the second smaller
        for epilogue, top_db, per_clip in ((_lib.EPI_WHISPER, -1.0, 1), (_lib.EPI_WHISPER, -1.0, 0),
                                           (_lib.EPI_DB, 80.0, 1), (_lib.EPI_NONE, -1.0, 1)):
            want, got = _both(plan, y, epilogue, top_db, per_clip)
            assert np.array_equal(want, got, equal_nan=True), (y.shape, epilogue, per_clip)


def test_static_entry_point_generic_kernel():
    """A framing served by the generic forward kernel (n_fft 1024) takes the same keys."""
    from mlx_audio_amd._runtime import get_plan
    from mlx_audio_amd.primitives.mel import mel_filterbank_np
    from mlx_audio_amd.primitives.windows import padded_window_np

    basis = mel_filterbank_np(48000, 1024, 64, 50.0, 14000.0, True, "slaney")
    plan = get_plan(torch.device("cuda", 0), 1024, 480, True, "constant",
                    padded_window_np("hann", 1024, 1024), mel_basis=basis, power=2.0)
    g = torch.Generator(device="cuda").manual_seed(11)
    for shape in ((6, 48000), (2, 9000)):
        y = torch.randn(shape, device="cuda", generator=g) * torch.logspace(
            -4, 0, shape[1], device="cuda")[None, :]
        want, got = _both(plan, y.contiguous(), _lib.EPI_DB, 30.0, 1)
        assert np.array_equal(want, got)


def test_static_entry_point_rejects_oversized_batches():
    plan = _whisper_plan(80)
    lib = _lib.lib()
    y = torch.zeros((1, 16000), device="cuda")
    out = torch.empty((1, 80, plan.num_frames(16000)), device="cuda")
    rc = lib.mia_melspectrogram_static(plan.handle, y.data_ptr(), 65536, 16000, _lib.EPI_WHISPER,
                                       1.0, 1e-10, 10.0, -1.0, 1, out.data_ptr(),
                                       ctypes.c_void_p(0))
    assert rc == _lib.MIA_ERR_INVALID
    assert b"65535" in lib.mia_last_error()
