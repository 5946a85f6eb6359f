"""Golden vectors (tests/golden/spectral_golden.npz, made by make_golden.py from
float64 torch / scipy / transformers computations on the reference's canonical
edge-case signals) against the CPU oracle — and, on a GPU, against the HIP
path through the C ABI.  Tolerances are the reference's own
(python/tests/parity/test_stft_parity.py: magnitude rtol 1e-4, round trip
atol 1e-4, filterbank atol 1e-6) and the 1e-4 log-mel bound of the headline
benchmark.
"""

import os

import numpy as np
import pytest

import oracle.spectral as o

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                         "spectral_golden.npz"), allow_pickle=False)
X = G["signals"]  # (9, 4000) float32


def _stft_close(got, want):
    scale = np.abs(want).max(axis=(1, 2), keepdims=True) + 1e-30
    assert (np.abs(got - want) / scale).max() < 2e-6
    np.testing.assert_allclose(np.abs(got), np.abs(want), rtol=1e-4, atol=1.5e-4)


# ------------------------------------------------------------------ CPU: oracle


def test_oracle_tables():
    for name in ("hann", "hamming", "blackman", "bartlett"):
        np.testing.assert_allclose(o.get_window(name, 400), G[f"window_{name}_400"], atol=1e-7)
        np.testing.assert_allclose(o.get_window(name, 33, False), G[f"window_{name}_33_sym"],
                                   atol=1e-7)
    for n_mels in (80, 128):
        np.testing.assert_allclose(o.mel_filterbank(16000, 400, n_mels, 0.0, 8000.0),
                                   G[f"melfb_16000_400_{n_mels}"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(o.mel_filterbank(48000, 1024, 64, 50.0, 14000.0, htk=True),
                               G["melfb_48000_1024_64_htk"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(o.dct_matrix(40, 128), G["dct_40x128_ortho"], atol=1e-7)
    np.testing.assert_allclose(o.power_to_db(G["db_known_in"].astype(np.float32)),
                               G["db_known_power"], atol=1e-5)
    np.testing.assert_allclose(o.amplitude_to_db(G["db_known_in"].astype(np.float32)),
                               G["db_known_amplitude"], atol=1e-5)


def test_oracle_stft_istft():
    _stft_close(o.stft(X, 400, 160), G["stft400"])
    _stft_close(o.stft(X, 512, 128, pad_mode="reflect"), G["stft512_reflect"])
    y = o.istft(G["stft512_reflect"], 128, length=X.shape[1])
    np.testing.assert_allclose(y[:, 8:-8], G["istft512"][:, 8:-8], atol=1e-4)
    np.testing.assert_allclose(y[:, 8:-8], X[:, 8:-8], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_oracle_whisper_logmel(n_mels):
    for i in range(X.shape[0]):  # per-clip maximum: one call per clip
        got = o.compute_log_mel_spectrogram(X[i], n_mels=n_mels)
        np.testing.assert_allclose(got, G[f"whisper_logmel_{n_mels}"][i], atol=1e-4,
                                   err_msg=str(G["signal_names"][i]))


def test_oracle_mfcc():
    for i in range(X.shape[0]):
        got = o.mfcc(X[i], sr=16000, n_mfcc=13, n_fft=512, hop_length=128, n_mels=40, fmax=8000.0)
        np.testing.assert_allclose(got, G["mfcc_512_40_13"][i], rtol=1e-4, atol=5e-3,
                                   err_msg=str(G["signal_names"][i]))


# ------------------------------------------------------------------ GPU: HIP path


@pytest.mark.gpu
def test_hip_stft_istft_golden(gpu):
    import torch

    from mlx_audio_amd.primitives import istft, stft

    x = torch.from_numpy(X).to(gpu)
    _stft_close(stft(x, 400, 160).cpu().numpy(), G["stft400"])
    S = stft(x, 512, 128, pad_mode="reflect")
    _stft_close(S.cpu().numpy(), G["stft512_reflect"])
    y = istft(torch.from_numpy(G["stft512_reflect"]).to(gpu), 128, length=X.shape[1]).cpu().numpy()
    np.testing.assert_allclose(y[:, 8:-8], G["istft512"][:, 8:-8], atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("n_mels", [80, 128])
def test_hip_whisper_logmel_golden(gpu, n_mels):
    import torch

    from mlx_audio_amd.frontends import compute_log_mel_spectrogram

    x = torch.from_numpy(X).to(gpu)
    got = compute_log_mel_spectrogram(x, n_mels=n_mels, per_clip=True).cpu().numpy()
    want = G[f"whisper_logmel_{n_mels}"]
    for i in range(X.shape[0]):
        np.testing.assert_allclose(got[i], want[i], atol=1e-4, err_msg=str(G["signal_names"][i]))
        one = compute_log_mel_spectrogram(x[i], n_mels=n_mels).cpu().numpy()
        np.testing.assert_allclose(one, want[i], atol=1e-4)


@pytest.mark.gpu
def test_hip_mfcc_golden(gpu):
    import torch

    from mlx_audio_amd.primitives import mfcc

    x = torch.from_numpy(X).to(gpu)
    got = mfcc(x, sr=16000, n_mfcc=13, n_fft=512, hop_length=128, n_mels=40, fmax=8000.0,
               per_clip=True).cpu().numpy()
    np.testing.assert_allclose(got, G["mfcc_512_40_13"], rtol=1e-4, atol=5e-3)


# ------------------------------------------------- whole chains from Hugging Face's feature code
# tests/golden/hf_features_golden.npz (make_golden_hf.py): transformers.audio_utils.spectrogram /
# mel_filter_bank / window_function - the routines behind WhisperFeatureExtractor and
# ClapFeatureExtractor - in float64, on a speech-like clip with a quiet stretch, ragged noise
# and a chirp.  OpenAI's Whisper front end pads by reflection where the reference pads with
# zeros (primitives/stft.py:143), so the Whisper vectors pin the chain with pad_mode="reflect".

H = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                         "hf_features_golden.npz"), allow_pickle=False)


def _whisper_epilogue_without_last_frame(mel):
    ls = np.log10(np.maximum(np.asarray(mel, dtype=np.float64), 1e-10))[..., :-1]
    return (np.maximum(ls, ls.max() - 8.0) + 4.0) / 4.0


@pytest.mark.parametrize("n_mels", [80, 128])
@pytest.mark.parametrize("clip", ["a", "b"])
def test_oracle_whisper_chain_against_hf(n_mels, clip):
    y = H[f"whisper_in_{clip}"]
    want = H[f"whisper{n_mels}_{clip}"]
    for dtype, tol in ((np.float64, 5e-6), (np.float32, 2e-5)):
        mel = o.melspectrogram(y, sr=16000, n_fft=400, hop_length=160, n_mels=n_mels, fmin=0.0,
                               fmax=8000.0, pad_mode="reflect", dtype=dtype)
        got = _whisper_epilogue_without_last_frame(mel)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= tol, (n_mels, clip, dtype)
    # interior frames do not see the padding: the reference's own (zero-padded) form agrees there
    own = o.compute_log_mel_spectrogram(y, n_mels=n_mels)[:, 2:want.shape[1] - 2]
    ref = want[:, 2:-2]
    clear = (own > own.min() + 1e-3) & (ref > ref.min() + 1e-3)  # above both floors
    assert clear.mean() > 0.5 and np.abs(own - ref)[clear].max() <= 2e-5


def test_oracle_clap_chain_against_hf():
    want = H["clap64"]
    for dtype, tol in ((np.float64, 2e-4), (np.float32, 5e-4)):  # dB
        got = np.asarray(o.clap_mel(H["clap_in"], dtype=dtype))
        assert got.shape == want.shape == (101, 64)
        assert np.abs(got - want).max() <= tol, dtype


@pytest.mark.gpu
@pytest.mark.parametrize("n_mels", [80, 128])
@pytest.mark.parametrize("clip", ["a", "b"])
def test_hip_whisper_chain_against_hf(gpu, n_mels, clip):
    """The n_fft = 400 / hop 160 kernel (reflect padding) against Hugging Face's Whisper features:
    its mel powers with the epilogue applied here, and its own fused log epilogue wherever the
    clip's maximum - which the kernel takes over all frames, OpenAI over all but the last - is the
    same."""
    import torch

    from mlx_audio_amd import _lib
    from mlx_audio_amd.primitives.mel import _mel_raw

    y = torch.from_numpy(H[f"whisper_in_{clip}"]).to(gpu)[None]
    want = H[f"whisper{n_mels}_{clip}"]
    args = (y, 16000, 400, 160, None, "hann", True, "reflect", 2.0, n_mels, 0.0, 8000.0, False, "slaney")
    mel = _mel_raw(*args).cpu().numpy()[0]
    assert np.abs(_whisper_epilogue_without_last_frame(mel) - want).max() <= 2e-5
    fused = _mel_raw(*args, epilogue=_lib.EPI_WHISPER, per_clip=True).cpu().numpy()[0]
    if fused.max() == fused[:, :-1].max():
        assert np.abs(fused[:, :-1] - want).max() <= 2e-5
    else:
        clear = want > want.min() + 1e-3
        assert np.abs(fused[:, :-1] - want)[clear].max() <= 2e-5


@pytest.mark.gpu
def test_hip_clap_chain_against_hf(gpu):
    import torch

    from mlx_audio_amd.frontends import clap_log_mel

    got = clap_log_mel(torch.from_numpy(H["clap_in"]).to(gpu)).cpu().numpy()
    assert got.shape == (101, 64)
    assert np.abs(got - H["clap64"]).max() <= 2e-3  # dB
