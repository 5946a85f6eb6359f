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


def test_htdemucs_batch64_round_trip_full_size(gpu):
    """BASELINE configs[2] at its stated size: 64 stereo tracks x 180 s @ 44.1 kHz = 128 rows of
    7 938 000 samples, n_fft 4096 / hop 1024: a 16.3 GB spectrum (2.03 G complex values = 4.07 G
    floats, past signed 32-bit offsets), round-trip SNR >= 100 dB, rows against the oracle,
    batch invariance."""
    from mlx_audio_amd.primitives import istft, stft

    gen = torch.Generator(device=gpu)
    gen.manual_seed(42)
    T = 44100 * 180
    rows = 128
    x = 0.1 * torch.randn((rows, T), generator=gen, device=gpu)
    S = stft(x, n_fft=4096, hop_length=1024, pad_mode="reflect")
    F = 1 + T // 1024
    assert tuple(S.shape) == (rows, 2049, F) and 2 * S.numel() > 2 ** 31
    y = istft(S, hop_length=1024, length=T)
    assert tuple(y.shape) == (rows, T)
    # SNR over the whole batch and of the worst row, accumulated in float64 row by row
    num = den = 0.0
    worst = 1e9
    for r in range(rows):
        xr, yr = x[r].double(), y[r].double()
        n_r, d_r = float((xr ** 2).sum()), float(((xr - yr) ** 2).sum())
        num, den = num + n_r, den + d_r
        worst = min(worst, 10 * np.log10(n_r / d_r))
    assert 10 * np.log10(num / den) >= 100.0 and worst >= 100.0
    # first / middle / last row against the float64 oracle (the last rows lie beyond 2^31 floats)
    for r in (0, 63, rows - 1):
        want = o.stft(x[r].cpu().numpy(), n_fft=4096, hop_length=1024, pad_mode="reflect",
                      dtype=np.float64)
        got = S[r].cpu().numpy()
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max(), r
    # a row of the batch equals the same row alone, bit for bit (forward and inverse)
    alone = stft(x[rows - 1], n_fft=4096, hop_length=1024, pad_mode="reflect")
    assert torch.equal(alone, S[rows - 1])
    assert torch.equal(istft(alone, hop_length=1024, length=T), y[rows - 1])
    del S, y, x
    torch.cuda.empty_cache()


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
    """BASELINE configs[3] per-GPU shard: 1024 clips of 10 s @ 48 kHz (batch 8192 over 8 GPUs),
    n_fft 1024 / hop 480, 64 mels."""
    from mlx_audio_amd.frontends import clap_log_mel

    gen = torch.Generator(device=gpu)
    gen.manual_seed(7)
    y = torch.randn((1024, 480000), generator=gen, device=gpu)
    out = clap_log_mel(y)
    assert tuple(out.shape) == (1024, 1001, 64)
    for b in (0, 511, 1023):
        ref = o.clap_mel(y[b].cpu().numpy())
        assert np.abs(out[b].cpu().numpy() - ref).max() <= 2e-3  # dB
    assert torch.equal(clap_log_mel(y[17]), out[17])


def test_mfcc_batch_against_oracle_spot(gpu):
    """BASELINE configs[4] per-clip shape (30 s @ 16 kHz, n_fft 2048 / hop 512, 128 mels, 40 coeffs)."""
    from mlx_audio_amd.primitives import mfcc

    gen = torch.Generator(device=gpu)
    gen.manual_seed(1000)
    n = 4096  # the per-GPU pool of the 10k-hour stream (7.9 GB of audio)
    y = torch.randn((n, 480000), generator=gen, device=gpu)
    out = mfcc(y, sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128, per_clip=True)
    assert tuple(out.shape) == (n, 40, 938)
    assert torch.isfinite(out).all()
    for b in (0, 2047, n - 1):
        ref = o.mfcc(y[b].cpu().numpy(), sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128)
        np.testing.assert_allclose(out[b].cpu().numpy(), ref, rtol=1e-5, atol=2e-3)
    alone = mfcc(y[9], sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128)
    assert torch.equal(alone, out[9])


def test_mfcc_more_rows_than_one_launch_takes(gpu):
    """A stream cut into more than 65 535 clips in one call (a launch's row limit): the call
    goes through in pieces, every clip as if it were alone."""
    from mlx_audio_amd.primitives import mfcc

    gen = torch.Generator(device=gpu)
    gen.manual_seed(5)
    n, T = 70000, 4096
    y = torch.randn((n, T), generator=gen, device=gpu)
    out = mfcc(y, sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128, per_clip=True)
    assert tuple(out.shape) == (n, 40, 9)
    for b in (0, 65534, 65535, 65536, n - 1):
        alone = mfcc(y[b], sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128)
        assert torch.equal(alone, out[b]), b
    ref = o.mfcc(y[n - 1].cpu().numpy(), sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128)
    np.testing.assert_allclose(out[n - 1].cpu().numpy(), ref, rtol=1e-5, atol=2e-3)
    with pytest.raises(ValueError, match="per_clip"):
        mfcc(y, sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128, per_clip=False)
