"""Builds libmia_hip.so (gfx950) next to the sources with hipcc.

Usage: python mlx_audio_amd/csrc/build.py [--force]
Each .hip file is compiled to an object only when it or a header is newer.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["capi.hip", "ops.hip", "spectral_fwd.hip", "spectral_lane.hip", "spectral_wave.hip", "spectral_inv.hip", "logmel400.hip", "resample.hip", "bigfft.hip", "features.hip", "onset.hip", "griffinlim.hip", "gate.hip", "normalize.hip", "stem.hip"]
HEADERS = ["fft_engine.h", "mia_internal.h", "w400_sweep_gen.inc", os.path.join(ROOT, "include", "mia.h")]
LIB = os.path.join(HERE, "libmia_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("MIA_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _mtime(path: str) -> float:
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    hdr_time = max(_mtime(h if os.path.isabs(h) else os.path.join(HERE, h)) for h in HEADERS)
    jobs = []
    objs = []
    for src in SOURCES:
        spath = os.path.join(HERE, src)
        opath = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(opath)
        if force or _mtime(opath) < max(_mtime(spath), hdr_time):
            jobs.append([hipcc, *FLAGS, "-c", spath, "-o", opath])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _mtime(LIB) < max(_mtime(o) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
