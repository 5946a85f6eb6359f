"""Generates w400_sweep_gen.inc: the mel sweep of the n_fft = 400 / hop = 160 kernel as
straight-line code for Whisper's two filterbanks (16 kHz, 0..8000 Hz, Slaney scale and
norm, 80 and 128 filters).

A triangular filterbank gives every FFT bin at most two non-zero weights: the rising edge
of filter s(k) and the falling edge of filter s(k) - 1.  For a fixed bank the per-wave
sweep over the power tile is therefore a fixed sequence of "load bin, multiply-add into
one or two accumulators, emit a finished filter" steps.  This script writes that sequence
out per wave with the weights as literals, so that the kernel carries no per-bin control
flow and no weight loads; banks other than these two keep the table-driven sweep.

The weights are computed here exactly as mia_mel_filterbank computes them (float64 slopes
with the 1e-10 regulariser, rounded to float32, then the Slaney scale applied to the
float32 value in float64 and rounded again).  The generated file also carries every
non-zero (filter, bin, bits) triple; mia_plan_create compares the caller's filterbank with
it bit for bit and only then selects the compiled sweep.

Usage: python gen_w400_sweep.py [--check]   (--check: fail if the file on disk is stale)
"""
from __future__ import annotations

import math
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "w400_sweep_gen.inc")

SR, N_FFT, FMIN, FMAX = 16000, 400, 0.0, 8000.0
BINS = N_FFT // 2 + 1
WAVES = 16
BANKS = (80, 128)

# Cost model of one wave's sweep, in issue slots: a bin costs its LDS read plus one slot
# per non-zero weight; an emit costs floor + log (two slots) + scale + the running extremes
# + select, lane rotation and its address, store and its offset (the line-aligned stores).
COST_READ, COST_FMA, COST_EMIT = 1.0, 1.0, float(os.environ.get("GEN_COST_EMIT", "7"))
LOADS_FIRST = os.environ.get("GEN_LDFIRST", "0") == "1"
# Waves 4 j .. 4 j + 3 are the j-th oldest on their SIMDs; the issue arbiter serves older
# waves first, so the younger ones get a smaller share of the sweep.
SHARE = tuple(float(v) for v in os.environ.get("GEN_SHARE", "1.10,1.05,0.95,0.90").split(","))


def _hz_to_mel(f: float) -> float:
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    if f < min_log_hz:
        return f / f_sp
    return min_log_mel + math.log(f / min_log_hz) / logstep


def _mel_to_hz(m: float) -> float:
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    if m < min_log_mel:
        return f_sp * m
    return min_log_hz * math.exp(logstep * (m - min_log_mel))


def _linspace(a: float, b: float, n: int) -> list[float]:
    step = (b - a) / (n - 1)
    v = [a + i * step for i in range(n)]
    v[-1] = b
    return v


def filterbank(n_mels: int) -> np.ndarray:
    """(n_mels, BINS) float32, the arithmetic of mia_mel_filterbank (slaney norm)."""
    freqs = _linspace(0.0, SR / 2.0, BINS)
    mels = _linspace(_hz_to_mel(FMIN), _hz_to_mel(FMAX), n_mels + 2)
    hz = [_mel_to_hz(m) for m in mels]
    fb = np.zeros((n_mels, BINS), dtype=np.float32)
    for m in range(n_mels):
        lo, ce, hi = hz[m], hz[m + 1], hz[m + 2]
        enorm = 2.0 / (hi - lo)
        for k in range(BINS):
            f = freqs[k]
            lower = (f - lo) / (ce - lo + 1e-10)
            upper = (hi - f) / (hi - ce + 1e-10)
            w = np.float32(max(0.0, min(lower, upper)))
            fb[m, k] = np.float32(float(w) * enorm)
    return fb


def segments(fb: np.ndarray):
    """seg[k], rising weight wr[k] (into filter seg[k]) and falling weight wf[k] (into
    filter seg[k] - 1) of every bin; raises if the bank is not of that form."""
    n_mels, bins = fb.shape
    seg = [0] * bins
    wr = [np.float32(0)] * bins
    wf = [np.float32(0)] * bins
    cur = 0
    for k in range(bins):
        nz = [m for m in range(n_mels) if fb[m, k] != 0]
        if len(nz) == 0:
            seg[k] = cur
        elif len(nz) == 1:
            s = nz[0] if nz[0] >= cur else nz[0] + 1
            assert s >= cur
            cur = s
            seg[k] = s
        else:
            assert len(nz) == 2 and nz[1] == nz[0] + 1 and nz[1] >= cur, (k, nz)
            cur = nz[1]
            seg[k] = cur
        s = seg[k]
        if s < n_mels:
            wr[k] = fb[s, k]
        if 1 <= s <= n_mels:
            wf[k] = fb[s - 1, k]
    for m in range(n_mels):
        for k in range(bins):
            s = seg[k]
            v = wr[k] if m == s else (wf[k] if m == s - 1 else np.float32(0))
            assert v == fb[m, k], (m, k)
    return seg, wr, wf


def wave_program(seg, wr, wf, m0: int, m1: int):
    """Steps of the wave that owns filters [m0, m1): ('bin', k, [(acc, w), ...]) and
    ('emit', acc or None)."""
    steps = []
    started = set()
    for s in range(m0, m1 + 1):
        for k in range(len(seg)):
            if seg[k] != s:
                continue
            terms = []
            if s > m0 and wf[k] != 0:
                terms.append((s - 1, wf[k]))
            if s < m1 and wr[k] != 0:
                terms.append((s, wr[k]))
            if terms:
                steps.append(("bin", k, terms))
        if s > m0:
            steps.append(("emit", s - 1))
    # mark first contributions
    out = []
    for st in steps:
        if st[0] == "bin":
            terms = []
            for acc, w in st[2]:
                terms.append((acc, w, acc not in started))
                started.add(acc)
            out.append(("bin", st[1], terms))
        else:
            out.append(("emit", st[1], st[1] in started))
    return out


def program_cost(prog) -> float:
    c = 0.0
    for st in prog:
        if st[0] == "bin":
            c += COST_READ + COST_FMA * len(st[2])
        else:
            c += COST_EMIT
    return c


def deal(seg, wr, wf, n_mels: int):
    """Contiguous runs of filters per wave with even (age-weighted) sweep cost: the
    smallest bound for which a greedy left-to-right deal places every filter."""

    def try_bound(bound):
        starts = [0] * (WAVES + 1)
        m = 0
        for w in range(WAVES):
            starts[w] = m
            bw = bound * SHARE[w // 4]
            while m < n_mels and program_cost(wave_program(seg, wr, wf, starts[w], m + 1)) <= bw:
                m += 1
        starts[WAVES] = m
        return starts if m == n_mels else None

    lo, hi = 0.0, 1e6
    for _ in range(50):
        mid = 0.5 * (lo + hi)
        if try_bound(mid) is not None:
            hi = mid
        else:
            lo = mid
    return try_bound(hi)


def fhex(w) -> str:
    return float(w).hex() + "f"


def bits(w) -> int:
    return struct.unpack("<I", struct.pack("<f", float(w)))[0]


def generate() -> str:
    L = []
    L.append("// Generated by gen_w400_sweep.py - do not edit; regenerate with")
    L.append("//   python mlx_audio_amd/csrc/gen_w400_sweep.py")
    L.append("// The mel sweep of the n_fft = 400 kernel as straight-line code for Whisper's 80- and")
    L.append("// 128-filter banks.  The including file defines, for a context object c:")
    L.append("//   SW_LD(x, k)       x = power of bin k of this lane's frame")
    L.append("//   SW_SYNC()         every SW_LD of the wave has been issued")
    L.append("//   SW_MUL(a, w, x)   a = w * x          (first contribution to accumulator a)")
    L.append("//   SW_FMA(a, w, x)   a = fma(w, x, a)")
    L.append("//   SW_EMIT(j, a)   output row j of the wave is complete with value a")
    L.append("//   SW_EMIT0(j)     ... with value 0 (a filter between two bins)")
    L.append("//   SW_END(n)       the wave's sweep is over; it emitted n rows")
    L.append("#pragma once")
    for n_mels in BANKS:
        fb = filterbank(n_mels)
        seg, wr, wf = segments(fb)
        starts = deal(seg, wr, wf, n_mels)
        nz = [(m, k) for m in range(n_mels) for k in range(BINS) if fb[m, k] != 0]
        L.append("")
        L.append(f"// ---- {n_mels} filters: {len(nz)} non-zero weights ----")
        L.append(f"#define W400_BANK{n_mels}_NNZ {len(nz)}")
        L.append(f"static const unsigned short kW400Bank{n_mels}Idx[W400_BANK{n_mels}_NNZ][2] = {{")
        for i in range(0, len(nz), 8):
            L.append("  " + " ".join(f"{{{m},{k}}}," for m, k in nz[i:i + 8]))
        L.append("};")
        L.append(f"static const unsigned kW400Bank{n_mels}Bits[W400_BANK{n_mels}_NNZ] = {{")
        for i in range(0, len(nz), 6):
            L.append("  " + " ".join(f"0x{bits(fb[m, k]):08x}u," for m, k in nz[i:i + 6]))
        L.append("};")
        L.append(f"#define W400_BANK{n_mels}_MAXROWS {max(starts[w + 1] - starts[w] for w in range(WAVES))}")
        L.append(f"// first filter of every wave (entry 16 = {n_mels})")
        L.append(f"__device__ static const int kW400Start{n_mels}[{WAVES + 1}] = {{"
                 + ", ".join(str(s) for s in starts) + "};")
        costs = []
        L.append("template <class Ctx>")
        L.append(f"__device__ __forceinline__ void w400_sweep_{n_mels}(int wv, Ctx& c) {{")
        L.append("  switch (wv) {")
        for w in range(WAVES):
            m0, m1 = starts[w], starts[w + 1]
            prog = wave_program(seg, wr, wf, m0, m1)
            costs.append(program_cost(prog))
            L.append(f"    case {w}: {{  // filters {m0}..{m1 - 1}, cost {costs[-1]:.0f}")
            accs = sorted({t[0] for st in prog if st[0] == "bin" for t in st[2]})
            bins_w = [st[1] for st in prog if st[0] == "bin"]
            if bins_w:
                L.append("      float " + ", ".join(f"x{i}" for i in range(len(bins_w))) + ";")
            if accs:
                L.append("      float " + ", ".join(f"a{a}" for a in accs) + ";")
            # every power value of the wave's bins first (the reads are then all in flight
            # together and the arithmetic waits for them in order), then the arithmetic
            if LOADS_FIRST:
                for i in range(0, len(bins_w), 6):
                    L.append("      " + " ".join(f"SW_LD(x{i + j}, {k})"
                                                 for j, k in enumerate(bins_w[i:i + 6])))
                L.append("      SW_SYNC()")
            ib = 0
            for st in prog:
                if st[0] == "bin":
                    if not LOADS_FIRST:
                        L.append(f"      SW_LD(x{ib}, {st[1]})")
                    ops = " ".join(
                        (f"SW_MUL(a{a}, {fhex(wt)}, x{ib})" if first
                         else f"SW_FMA(a{a}, {fhex(wt)}, x{ib})")
                        for a, wt, first in st[2])
                    L.append(f"      {ops}")
                    ib += 1
                else:
                    j = st[1] - m0
                    L.append(f"      SW_EMIT({j}, a{st[1]})" if st[2] else f"      SW_EMIT0({j})")
            L.append(f"      SW_END({m1 - m0})")
            L.append("    } break;")
        L.append("    default: break;")
        L.append("  }")
        L.append("}")
        L.append(f"// sweep cost per wave: {' '.join(f'{c:.0f}' for c in costs)}")
    return "\n".join(L) + "\n"


def main() -> int:
    text = generate()
    if "--check" in sys.argv:
        on_disk = open(OUT).read() if os.path.exists(OUT) else ""
        if on_disk != text:
            print("w400_sweep_gen.inc is stale: run gen_w400_sweep.py", file=sys.stderr)
            return 1
        return 0
    with open(OUT, "w") as f:
        f.write(text)
    print(f"wrote {OUT}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
