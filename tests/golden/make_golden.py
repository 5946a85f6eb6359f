"""Generates tests/golden/spectral_golden.npz.

The reference cannot run offline (its array library is absent) and stores no
numbers for this path, so these vectors come from implementations that are
independent of both the oracle and the HIP kernels, evaluated in float64:
CPU torch.stft / torch.istft, scipy.signal.get_window, scipy.fft.dct,
transformers.audio_utils.mel_filter_bank, and closed forms.  Inputs are the
reference's canonical edge-case signals (python/tests/generate_edge_case_fixtures.py:38-136:
short sine, silence, clipped sine, DC offset, impulse, noise, all-ones, tiny
amplitude, alternating +-1) plus seeded noise.

    python tests/golden/make_golden.py
"""

import os

import numpy as np
import scipy.fft
import scipy.signal
import torch
from transformers.audio_utils import mel_filter_bank

HERE = os.path.dirname(os.path.abspath(__file__))


def edge_signals(sr=16000, n=4000):
    t = np.arange(n) / sr
    rng = np.random.default_rng(2024)
    imp = np.zeros(n)
    imp[n // 2] = 1.0
    return {
        "sine440": np.sin(2 * np.pi * 440 * t),
        "silence": np.zeros(n),
        "clipped": np.clip(1.5 * np.sin(2 * np.pi * 440 * t), -1, 1),
        "dc": 0.5 + 0.5 * np.sin(2 * np.pi * 440 * t),
        "impulse": imp,
        "noise": 0.5 * rng.standard_normal(n),
        "ones": np.ones(n),
        "tiny": 1e-6 * np.sin(2 * np.pi * 440 * t),
        "alternating": np.where(np.arange(n) % 2 == 0, 1.0, -1.0),
    }


def t_stft(x, n_fft, hop, pad_mode, win):
    return torch.stft(torch.from_numpy(x), n_fft, hop_length=hop, win_length=n_fft,
                      window=torch.from_numpy(win), center=True, pad_mode=pad_mode,
                      return_complex=True).numpy()


def main():
    out = {}
    sig = edge_signals()
    names = sorted(sig)
    x = np.stack([sig[k] for k in names]).astype(np.float32).astype(np.float64)
    out["signal_names"] = np.array(names)
    out["signals"] = x.astype(np.float32)

    # Whisper front-end pieces: n_fft 400 / hop 160, hann, zero padding
    hann400 = scipy.signal.get_window("hann", 400, fftbins=True)
    S = t_stft(x, 400, 160, "constant", hann400)
    out["stft400"] = S.astype(np.complex64)
    for n_mels in (80, 128):
        fb = mel_filter_bank(201, n_mels, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
        out[f"melfb_16000_400_{n_mels}"] = fb
        mel = fb @ np.abs(S) ** 2
        out[f"mel400_{n_mels}"] = mel
        lm = np.log10(np.maximum(mel, 1e-10))
        lm = np.maximum(lm, lm.max(axis=(1, 2), keepdims=True) - 8.0)  # per clip
        out[f"whisper_logmel_{n_mels}"] = (lm + 4.0) / 4.0

    # reflect-padded STFT + inverse, n_fft 512 / hop 128
    hann512 = scipy.signal.get_window("hann", 512, fftbins=True)
    S512 = t_stft(x, 512, 128, "reflect", hann512)
    out["stft512_reflect"] = S512.astype(np.complex64)
    out["istft512"] = torch.istft(torch.from_numpy(S512), 512, hop_length=128,
                                  window=torch.from_numpy(hann512), center=True,
                                  length=x.shape[1]).numpy()

    # MFCC: 40 mels, 13 coefficients, top_db 80 per clip
    fb40 = mel_filter_bank(257, 40, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
    Sc = t_stft(x, 512, 128, "constant", hann512)
    db = 10.0 * np.log10(np.maximum(fb40 @ np.abs(Sc) ** 2, 1e-10))
    db = np.maximum(db, db.max(axis=(1, 2), keepdims=True) - 80.0)
    out["mel_db_512_40"] = db
    out["mfcc_512_40_13"] = scipy.fft.dct(db, type=2, axis=1, norm="ortho")[:, :13]

    # tables
    for name, sname in (("hann", "hann"), ("hamming", "hamming"), ("blackman", "blackman"),
                        ("bartlett", "bartlett")):
        out[f"window_{name}_400"] = scipy.signal.get_window(sname, 400, fftbins=True)
        out[f"window_{name}_33_sym"] = scipy.signal.get_window(sname, 33, fftbins=False)
    out["melfb_48000_1024_64_htk"] = mel_filter_bank(513, 64, 50.0, 14000.0, 48000, norm="slaney",
                                                     mel_scale="htk").T
    out["dct_40x128_ortho"] = scipy.fft.dct(np.eye(128), type=2, axis=0, norm="ortho")[:40]
    out["db_known_in"] = np.array([1.0, 10.0, 100.0])
    out["db_known_power"] = np.array([0.0, 10.0, 20.0])
    out["db_known_amplitude"] = np.array([0.0, 20.0, 40.0])

    path = os.path.join(HERE, "spectral_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
