"""Generates tests/golden/hf_features_golden.npz: whole front-end chains from an implementation
that is independent of the oracle and of the HIP kernels - Hugging Face transformers'
NumPy feature code (transformers.audio_utils.spectrogram / mel_filter_bank / window_function, the
routines behind WhisperFeatureExtractor and ClapFeatureExtractor), evaluated in float64.

  * whisper80 / whisper128: OpenAI Whisper's log-mel as WhisperFeatureExtractor computes it
    (hann 400 / hop 160, reflect padding, slaney filterbank 0..8 kHz, last frame dropped,
    log10 floored 8 below the clip's maximum, (x + 4) / 4).  The reference pads with zeros
    (primitives/stft.py:143), so its edge frames differ from OpenAI's; the vectors therefore pin
    the chain with pad_mode="reflect" - same framing, window, transform, filterbank and epilogue.
  * clap64: the dB mel features the reference takes from librosa (models/clap/feature_extractor.py:
    193-212: n_fft 1024 / hop 480, 64 htk-spaced slaney-normalised filters 50..14000 Hz, zero
    padding, 10 log10(max(S, 1e-10)), no top_db), frames first.

    python tests/golden/make_golden_hf.py
"""

import os

import numpy as np
from transformers.audio_utils import mel_filter_bank, spectrogram, window_function

HERE = os.path.dirname(os.path.abspath(__file__))


def whisper_hf(y, n_mels):
    fb = mel_filter_bank(num_frequency_bins=201, num_mel_filters=n_mels, min_frequency=0.0,
                         max_frequency=8000.0, sampling_rate=16000, norm="slaney", mel_scale="slaney")
    ls = spectrogram(y.astype(np.float64), window_function(400, "hann"), frame_length=400, hop_length=160,
                     power=2.0, mel_filters=fb, log_mel="log10")[:, :-1]
    ls = np.maximum(ls, ls.max() - 8.0)
    return (ls + 4.0) / 4.0


def clap_hf(y):
    fb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=64, min_frequency=50.0,
                         max_frequency=14000.0, sampling_rate=48000, norm="slaney", mel_scale="htk")
    db = spectrogram(y.astype(np.float64), window_function(1024, "hann"), frame_length=1024, hop_length=480,
                     power=2.0, center=True, pad_mode="constant", mel_filters=fb, log_mel="dB",
                     reference=1.0, min_value=1e-10, db_range=None)
    return db.T


def signals():
    rng = np.random.default_rng(77)
    t = np.arange(24000) / 16000.0
    speechlike = (0.3 * np.sin(2 * np.pi * 220 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) +
                  0.02 * rng.standard_normal(t.size))
    speechlike[6000:9000] *= 1e-3            # a quiet stretch that the floor touches
    noise = 0.1 * rng.standard_normal(16000 + 77)   # ragged length
    t48 = np.arange(48000) / 48000.0
    chirp = 0.5 * np.sin(2 * np.pi * (200 + 6000 * t48) * t48) + 0.01 * rng.standard_normal(t48.size)
    return (speechlike.astype(np.float32), noise.astype(np.float32), chirp.astype(np.float32))


def main():
    a, b, c = signals()
    out = {"whisper_in_a": a, "whisper_in_b": b, "clap_in": c}
    for n in (80, 128):
        out[f"whisper{n}_a"] = whisper_hf(a, n)
        out[f"whisper{n}_b"] = whisper_hf(b, n)
    out["clap64"] = clap_hf(c)
    np.savez_compressed(os.path.join(HERE, "hf_features_golden.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
