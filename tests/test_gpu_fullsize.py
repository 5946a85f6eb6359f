"""Parity at the benchmark's full sizes through size-independent properties.

The CPU oracle cannot process 1024 x 30 s in seconds, so the full-size runs
are checked by (a) spot clips against the oracle, (b) batch-invariance: a clip
inside the batch gives bit-identical features to the same clip alone, (c) the
silence / scaling laws of the log-mel map, (d) linearity of the STFT, and
(e) the STFT -> iSTFT round trip (SNR >= 100 dB) on long multi-row inputs.
"""

import numpy as np
import pytest
import torch

import oracle.spectral as o

pytestmark = pytest.mark.gpu


def _snr_db(x: torch.Tensor, y: torch.Tensor) -> float:
    x64, y64 = x.double(), y.double()
    return float(10 * torch.log10((x64 ** 2).sum() / ((x64 - y64) ** 2).sum()))


def test_whisper128_full_batch(gpu):
    """BASELINE configs[1]: 1024 x 30 s -> (1024, 128, 3001)."""
    from mlx_audio_amd.frontends import compute_log_mel_spectrogram

    gen = torch.Generator(device=gpu)
    gen.manual_seed(1234)
    y = torch.randn((1024, 480000), generator=gen, device=gpu)
    y[5] = 0.0                      # silence
    y[6] = y[7] * 1e-3              # a scaled copy: power down by 1e-6 -> log10 shifts by -6
    out = compute_log_mel_spectrogram(y, n_mels=128, per_clip=True)
    assert tuple(out.shape) == (1024, 128, 3001)
    assert torch.isfinite(out).all()
    # (a) spot clips against the float32 oracle, first / middle / last clip of the batch
    for b in (0, 511, 1023):
        ref = o.compute_log_mel_spectrogram(y[b].cpu().numpy(), n_mels=128)
        assert np.abs(out[b].cpu().numpy() - ref).max() <= 1e-4, b
    # (b) batch invariance, bit for bit
    for b in (0, 300, 1023):
        alone = compute_log_mel_spectrogram(y[b], n_mels=128)
        assert torch.equal(alone, out[b]), b
    # (c) silence and the scaling law
    assert (out[5] + 1.5).abs().max() <= 1e-6
    away = out[6] > -1.4  # the rare tiny values of the scaled clip sit on the absolute 1e-10 floor
    assert away.float().mean() > 0.999
    assert (out[6] - (out[7] - 6.0 / 4.0))[away].abs().max() <= 2e-5
    # every clip of white noise has the same statistics: per-clip means agree closely
    means = out.mean(dim=(1, 2))
    keep = torch.ones(1024, dtype=torch.bool, device=gpu)
    keep[5] = keep[6] = False
    assert (means[keep] - means[keep].mean()).abs().max() < 5e-3


def test_whisper80_reference_cpu_case(gpu):
    """BASELINE configs[0]: 16 x 30 s, 80 mels, one call per clip, all against the oracle."""
    from mlx_audio_amd.frontends import compute_log_mel_spectrogram

    rng = np.random.default_rng(42)
    x = rng.standard_normal((16, 480000), dtype=np.float32)
    got = compute_log_mel_spectrogram(torch.from_numpy(x).to(gpu), n_mels=80, per_clip=True)
    got = got.cpu().numpy()
    for b in range(16):
        ref = o.compute_log_mel_spectrogram(x[b], n_mels=80)
        assert np.abs(got[b] - ref).max() <= 1e-4, b


def test_stft_linearity_and_round_trip_htdemucs_size(gpu):
    """BASELINE configs[2] shape per row (180 s @ 44.1 kHz, n_fft 4096 / hop 1024), 16 rows."""
    from mlx_audio_amd.primitives import istft, stft

    gen = torch.Generator(device=gpu)
    gen.manual_seed(42)
    T = 44100 * 180
    x = 0.1 * torch.randn((16, T), generator=gen, device=gpu)
    S = stft(x, n_fft=4096, hop_length=1024, pad_mode="reflect")
    assert tuple(S.shape) == (16, 2049, 1 + T // 1024)
    # (d) linearity: stft(a x0 + b x1) = a stft(x0) + b stft(x1)
    mix = stft(0.5 * x[0] - 2.0 * x[1], n_fft=4096, hop_length=1024, pad_mode="reflect")
    lin = 0.5 * S[0] - 2.0 * S[1]
    assert (mix - lin).abs().max() <= 2e-6 * float(lin.abs().max())
    # (e) round trip
    y = istft(S, hop_length=1024, length=T)
    assert tuple(y.shape) == (16, T)
    assert _snr_db(x, y) >= 100.0
    # a row of the batch equals the same row alone, bit for bit
    assert torch.equal(stft(x[3], n_fft=4096, hop_length=1024, pad_mode="reflect"), S[3])


def test_clap_batch_against_oracle_spot(gpu):
    """BASELINE configs[3] per-GPU shape (10 s @ 48 kHz, n_fft 1024 / hop 480, 64 mels), 256 clips."""
    from mlx_audio_amd.frontends import clap_log_mel

    gen = torch.Generator(device=gpu)
    gen.manual_seed(7)
    y = torch.randn((256, 480000), generator=gen, device=gpu)
    out = clap_log_mel(y)
    assert tuple(out.shape) == (256, 1001, 64)
    for b in (0, 255):
        ref = o.clap_mel(y[b].cpu().numpy())
        assert np.abs(out[b].cpu().numpy() - ref).max() <= 2e-3  # dB
    assert torch.equal(clap_log_mel(y[17]), out[17])


def test_mfcc_batch_against_oracle_spot(gpu):
    """BASELINE configs[4] per-clip shape (30 s @ 16 kHz, n_fft 2048 / hop 512, 128 mels, 40 coeffs)."""
    from mlx_audio_amd.primitives import mfcc

    gen = torch.Generator(device=gpu)
    gen.manual_seed(1000)
    y = torch.randn((128, 480000), generator=gen, device=gpu)
    out = mfcc(y, sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128, per_clip=True)
    assert tuple(out.shape) == (128, 40, 938)
    for b in (0, 127):
        ref = o.mfcc(y[b].cpu().numpy(), sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128)
        np.testing.assert_allclose(out[b].cpu().numpy(), ref, rtol=1e-5, atol=2e-3)
    alone = mfcc(y[9], sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128)
    assert torch.equal(alone, out[9])
