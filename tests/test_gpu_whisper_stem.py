"""Whisper encoder stem on the GPU (through the C ABI) against the oracle.

Tolerances.  precision "f32" (the default for float32 weights: float32 operands on the float32
matrix instruction, the arithmetic of the reference's nn.Conv1d on a float32 checkpoint) must
agree with the plain oracle (float64 arithmetic on the float32 weights) to 1e-5 of the output's
largest magnitude.  precision "f16": the kernels round the log-mel input, the weights and the intermediate activation
to float16 and accumulate in float32.  Against the oracle run on the SAME float16-rounded
weights and input (float64 arithmetic, intermediate rounded to float16 as the kernel stores it)
the float32 output must agree to 2e-3 of the output's largest magnitude (accumulation order and
the 1.5e-7 erfc polynomial only); against the plain float32 oracle to 1e-2 (float16 operands)."""
import numpy as np
import pytest
import torch

from mlx_audio_amd import _lib
from mlx_audio_amd.frontends import WhisperEncoderStem, compute_log_mel_spectrogram, sinusoids
from oracle import whisper_stem as ws

pytestmark = pytest.mark.gpu


def _weights(n_mels, n_state, seed):
    rng = np.random.default_rng(seed)
    w1 = (rng.standard_normal((n_state, 3, n_mels)) / np.sqrt(3 * n_mels)).astype(np.float32)
    b1 = (rng.standard_normal(n_state) * 0.1).astype(np.float32)
    w2 = (rng.standard_normal((n_state, 3, n_state)) / np.sqrt(3 * n_state)).astype(np.float32)
    b2 = (rng.standard_normal(n_state) * 0.1).astype(np.float32)
    return w1, b1, w2, b2


def _oracle_f16_model(mel, w1, b1, w2, b2, pos):
    """The oracle with the kernel's roundings: float16 input / weights / intermediate."""
    h = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float64)
    x = h(mel).transpose(0, 2, 1)
    a1 = ws.gelu(ws.conv1d_channels_last(x, h(w1), b1.astype(np.float64), 1, 1))
    a1 = a1.astype(np.float16).astype(np.float64)
    a2 = ws.gelu(ws.conv1d_channels_last(a1, h(w2), b2.astype(np.float64), 2, 1))
    if pos is not None:
        a2 = a2 + pos[None, :a2.shape[1], :].astype(np.float64)
    return a2


@pytest.mark.parametrize("n_mels,n_state,B,T", [(128, 1280, 2, 300), (80, 384, 3, 257), (128, 256, 1, 3000),
                                                (80, 128, 5, 1), (80, 128, 2, 2), (64, 512, 1, 130)])
@pytest.mark.parametrize("with_pos", [True, False])
def test_stem_matches_oracle(n_mels, n_state, B, T, with_pos):
    w1, b1, w2, b2 = _weights(n_mels, n_state, n_mels + n_state + T)
    rng = np.random.default_rng(T)
    mel = (rng.standard_normal((B, n_mels, T)) * 0.5).astype(np.float32)
    t_out = (T - 1) // 2 + 1
    pos = sinusoids(t_out + 2, n_state).numpy() if with_pos else None
    stem = WhisperEncoderStem(w1, b1, w2, b2, pos, precision="f16")
    got32 = stem(torch.from_numpy(mel).cuda(), out_dtype=torch.float32).cpu().numpy()
    got16 = stem(torch.from_numpy(mel).cuda(), out_dtype=torch.float16).float().cpu().numpy()
    assert got32.shape == (B, t_out, n_state)
    want = _oracle_f16_model(mel, w1, b1, w2, b2, pos)
    scale = np.abs(want).max()
    assert np.abs(got32 - want).max() <= 2e-3 * scale, np.abs(got32 - want).max() / scale
    assert np.abs(got16 - want).max() <= 2e-3 * scale + 1e-3 * scale  # + float16 output rounding
    plain = ws.encoder_stem(mel, w1, b1, w2, b2, pos, dtype=np.float64)
    assert np.abs(got32 - plain).max() <= 1e-2 * np.abs(plain).max()


@pytest.mark.parametrize("n_mels,n_state,B,T", [(128, 1280, 2, 300), (80, 384, 3, 257), (128, 256, 1, 3000),
                                                (80, 128, 5, 1), (80, 128, 2, 2), (64, 512, 1, 130)])
@pytest.mark.parametrize("with_pos", [True, False])
def test_float32_weights_run_in_float32(n_mels, n_state, B, T, with_pos):
    """Float32 weights are not narrowed unless the caller asks: the default precision for them
    is "f32", and its result agrees with the plain oracle to float32 round-off."""
    w1, b1, w2, b2 = _weights(n_mels, n_state, n_mels + n_state + T)
    rng = np.random.default_rng(T)
    mel = (rng.standard_normal((B, n_mels, T)) * 0.5).astype(np.float32)
    t_out = (T - 1) // 2 + 1
    pos = sinusoids(t_out + 2, n_state).numpy() if with_pos else None
    stem = WhisperEncoderStem(w1, b1, w2, b2, pos)
    assert stem.precision == "f32"
    got = stem(torch.from_numpy(mel).cuda())
    assert got.dtype == torch.float32 and tuple(got.shape) == (B, t_out, n_state)
    got = got.cpu().numpy()
    plain = ws.encoder_stem(mel, w1, b1, w2, b2, pos, dtype=np.float64)
    assert np.abs(got - plain).max() <= 1e-5 * np.abs(plain).max(), np.abs(got - plain).max() / np.abs(plain).max()
    # float16 output of the same arithmetic: only the final rounding is added
    got16 = stem(torch.from_numpy(mel).cuda(), out_dtype=torch.float16).float().cpu().numpy()
    assert np.abs(got16 - plain).max() <= 1e-3 * np.abs(plain).max()
    # float16 tensors select the float16 arithmetic by themselves
    half = WhisperEncoderStem(torch.from_numpy(w1).half(), b1, torch.from_numpy(w2).half(), b2, pos)
    assert half.precision == "f16" and half(torch.from_numpy(mel).cuda()).dtype == torch.float16
    with pytest.raises(ValueError):
        WhisperEncoderStem(w1, b1, w2, b2, pos, precision="bf16")


def test_float32_batch_invariance_and_tiles():
    """precision "f32": a clip's result does not depend on its neighbours, and the 256 x 256 tile
    of large batches gives the bits of the 128 x 128 tile (same order of the sum over k)."""
    w1, b1, w2, b2 = _weights(128, 1280, 21)
    stem = WhisperEncoderStem(w1, b1, w2, b2, sinusoids(1500, 1280))
    g = torch.Generator(device="cuda").manual_seed(22)
    mel = 0.5 * torch.randn((40, 128, 3000), device="cuda", generator=g)
    full = stem(mel)
    for i in (0, 23, 39):
        assert torch.equal(full[i:i + 1], stem(mel[i:i + 1])), i


def test_stem_on_log_mel_and_batch_invariance():
    """From audio: the log-mel of the fused front-end feeds the stem; a clip's result does not
    depend on its neighbours in the batch (bit for bit)."""
    w1, b1, w2, b2 = _weights(128, 1280, 5)
    pos = sinusoids(1500, 1280)
    stem = WhisperEncoderStem(w1, b1, w2, b2, pos, precision="f16")
    g = torch.Generator(device="cuda").manual_seed(3)
    y = torch.randn((5, 48000), device="cuda", generator=g)
    mel = compute_log_mel_spectrogram(y, n_mels=128, per_clip=True)
    full = stem(mel)
    one = stem(mel[3:4])
    assert torch.equal(full[3:4], one)
    want = ws.encoder_stem(mel[3:4].cpu().numpy(), w1, b1, w2, b2, pos.numpy(), dtype=np.float64)
    assert np.abs(one.float().cpu().numpy() - want).max() <= 1e-2 * np.abs(want).max()


def test_stem_argument_checks():
    w1, b1, w2, b2 = _weights(80, 128, 1)
    with pytest.raises(ValueError):
        WhisperEncoderStem(w1, b1, w2[:, :2], b2)
    with pytest.raises(NotImplementedError):
        WhisperEncoderStem(w1[:100], b1[:100], w2[:100, :, :100], b2[:100])
    stem = WhisperEncoderStem(w1, b1, w2, b2, sinusoids(4, 128))
    with pytest.raises(ValueError):
        stem(torch.zeros((1, 64, 10)))
    with pytest.raises(ValueError):
        stem(torch.zeros((1, 80, 100)))  # more output frames than positions
    lib = _lib.lib()
    assert lib.mia_whisper_stem_workspace_bytes(0, 10, 80, 128, _lib.STEM_F32) == 0
    assert lib.mia_whisper_stem_workspace_bytes(4, 10, 80, 128, 7) == 0   # unknown precision
    assert lib.mia_whisper_stem_weight_bytes(80, 128, _lib.STEM_F32) > lib.mia_whisper_stem_weight_bytes(80, 128, _lib.STEM_F16)
    assert stem(torch.zeros((0, 80, 6))).shape == (0, 3, 128)


def test_large_batch_tile_equals_small_batch_tile():
    """Batches large enough to fill the chip take the 256 x 256 tile, small ones the 128 x 128
    tile; every output element sums its products in the same order in both, so the results are
    equal bit for bit.  One clip is also checked against the oracle."""
    w1, b1, w2, b2 = _weights(128, 1280, 11)
    pos = sinusoids(1500, 1280)
    stem = WhisperEncoderStem(w1, b1, w2, b2, pos, precision="f16")
    g = torch.Generator(device="cuda").manual_seed(12)
    mel = 0.5 * torch.randn((40, 128, 3000), device="cuda", generator=g)
    for dt in (torch.float16, torch.float32):
        full = stem(mel, out_dtype=dt)
        for i in (0, 17, 39):
            assert torch.equal(full[i:i + 1], stem(mel[i:i + 1], out_dtype=dt)), (dt, i)
    want = _oracle_f16_model(mel[17:18].cpu().numpy(), w1, b1, w2, b2, pos.numpy())
    got = full[17:18].cpu().numpy()
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max()
