"""Which kernel sources a bench workload runs, and their git blob hashes.

`profiles/pmc_traffic.json` holds HBM byte counts measured with rocprofv3 counter passes.
Each entry records the blob hash of every source file its kernels are built from; the bench
line reports an entry's traffic only while those hashes still match the tree, so a kernel
edited after its last counter pass reports `traffic: null` instead of a stale figure.
"""
from __future__ import annotations

import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = "mlx_audio_amd/csrc"
TRAFFIC_FILE = "profiles/pmc_traffic.json"

_COMMON = ["fft_engine.h", "mia_internal.h"]
_FWD = ["spectral_fwd.hip", "ops.hip"] + _COMMON
# source files (under mlx_audio_amd/csrc) whose kernels one step of the workload launches
WORKLOAD_SOURCES = {
    "whisper128": ["logmel400.hip", "w400_sweep_gen.inc", "ops.hip"] + _COMMON,
    "clap64": _FWD,
    "mfcc40": _FWD,
    "htdemucs": ["spectral_fwd.hip", "spectral_inv.hip"] + _COMMON,
    "resample": ["resample.hip"],
    "resample_fft": ["bigfft.hip"] + _COMMON,
    "features": ["features.hip"],
    "features_all": ["features.hip"],
    "contrast": ["features.hip"],
    "features_y": ["spectral_fwd.hip", "features.hip"] + _COMMON,
    "onset": ["onset.hip"],
    "griffinlim": ["griffinlim.hip", "spectral_fwd.hip", "spectral_inv.hip", "ops.hip"] + _COMMON,
    "spectral_gate": ["gate.hip", "spectral_fwd.hip", "spectral_inv.hip", "ops.hip"] + _COMMON,
    "demucs_front": ["normalize.hip", "spectral_fwd.hip", "ops.hip"] + _COMMON,
    "whisper_stem": ["stem.hip"],
    "whisper_stem_f16": ["stem.hip"],
}


def blob_hash(path: str) -> str:
    """The hash `git hash-object` gives the file (sha1 of "blob <size>\\0" + content)."""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_hashes(workload: str, root: str = ROOT) -> dict:
    return {f"{CSRC}/{name}": blob_hash(os.path.join(root, CSRC, name))
            for name in WORKLOAD_SOURCES.get(workload, [])}


def measured_traffic(workload: str, clips: int, root: str = ROOT):
    """(hbm_bytes_per_step or None, traffic_source dict) for one workload and size.

    The figure is reported only when the entry names the source files it was measured on
    and every one of them still has the recorded blob hash."""
    src = {"file": TRAFFIC_FILE, "entry": f"{workload}:{clips}"}
    try:
        table = json.load(open(os.path.join(root, TRAFFIC_FILE)))
    except (OSError, ValueError):
        src["status"] = "no traffic table"
        return None, src
    entry = table.get(src["entry"])
    if entry is None:
        src["status"] = "no counter pass for this workload and size"
        return None, src
    recorded = entry.get("kernel_sources")
    if not recorded:
        src["status"] = "entry carries no source hashes"
        return None, src
    src["kernel_sources"] = recorded
    src["profile"] = entry.get("profile")
    changed = []
    for path, want in sorted(recorded.items()):
        try:
            have = blob_hash(os.path.join(root, path))
        except OSError:
            have = None
        if have != want:
            changed.append(path)
    if changed:
        src["status"] = "stale: changed since the counter pass: " + ", ".join(changed)
        return None, src
    src["status"] = "sources match the counter pass"
    return entry.get("hbm_bytes_per_step"), src
