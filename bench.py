"""Headline benchmark: Whisper 128-mel log-mel audio-hours/s on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

At N = 1 the default run also times, after the headline's timed region and outside its number,
the other BASELINE configurations (clap64, mfcc40, htdemucs) for a few steps each and attaches
them as "secondary"; "roofline.traffic" is a committed counter measurement that is reported
only while the kernel sources it was taken on are unchanged ("roofline.traffic_source").

One step = one pass of the hot path over one batch of synthetic clips that is
already resident in HBM.  With N > 1 (launched by torch.distributed.run, one
rank per GPU) every rank processes its own batch — clips shard with no
data-path collective — and the aggregate is reported (weak scaling).

Workloads (config.workload):
  whisper128  [default] 1024 x 30 s @16 kHz -> (1024, 128, 3001) log-mel
  clap64      1024 x 10 s @48 kHz -> (1024, 1001, 64) dB mel
  mfcc40      4096 x 30 s @16 kHz (the per-GPU pool of a stream), n_fft 2048 / hop 512 / 128 mels -> 40 MFCC
  htdemucs    64 stereo x 180 s @44.1 kHz STFT+iSTFT round trip, n_fft 4096 / hop 1024
  resample    512 x 30 s @44.1 kHz -> 16 kHz, polyphase (up 160 / down 441)
  resample_fft 64 x 30 s @44.1 kHz -> 16 kHz, Fourier method (the reference's default)
  features    centroid + bandwidth + roll-off + flatness of a (512, 1025, 938) spectrogram
  contrast    spectral contrast of the same spectrogram
  features_y  the four statistics from 512 x 30 s waveforms in one launch (fused epilogue)
  onset       onset strength of 4096 mel spectrograms (128 x 938)
  griffinlim  8 Griffin-Lim iterations on 64 x 10 s @22.05 kHz, n_fft 2048 / hop 512
  spectral_gate  spectral gating of 64 x 10 s @22.05 kHz
  demucs_front   HTDemucs spectrogram + input normalisation, 16 stereo x 60 s @44.1 kHz
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def make_workload(name: str, clips: int, device, seed: int):
    """Returns (step_fn, units_per_step, unit_name, algorithmic_bytes, description, metric)."""
    import torch

    from mlx_audio_amd.frontends import clap_log_mel, compute_log_mel_spectrogram
    from mlx_audio_amd.primitives import istft, mfcc, resample, resample_poly, stft

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    if name == "whisper128":
        T = 480000
        y = torch.randn((clips, T), generator=gen, device=device, dtype=torch.float32)
        frames = 3001
        abytes = clips * (T * 4 + 128 * frames * 4)
        return (lambda: compute_log_mel_spectrogram(y, n_mels=128, per_clip=True),
                clips * 30.0 / 3600.0, abytes,
                f"whisper-large-v3 128-mel log-mel, {clips} x 30 s clips @16 kHz, n_fft 400 hop 160",
                "log-mel audio-hours/sec (Whisper 128-mel)")
    if name == "clap64":
        T = 480000
        y = torch.randn((clips, T), generator=gen, device=device, dtype=torch.float32)
        abytes = clips * (T * 4 + 64 * 1001 * 4)
        return (lambda: clap_log_mel(y), clips * 10.0 / 3600.0, abytes,
                f"CLAP 64-mel dB features, {clips} x 10 s clips @48 kHz, n_fft 1024 hop 480",
                "CLAP 64-mel audio-hours/sec")
    if name == "mfcc40":
        T = 480000
        y = torch.randn((clips, T), generator=gen, device=device, dtype=torch.float32)
        abytes = clips * (T * 4 + 40 * 938 * 4)
        return (lambda: mfcc(y, sr=16000, n_mfcc=40, n_fft=2048, hop_length=512, n_mels=128,
                             per_clip=True),
                clips * 30.0 / 3600.0, abytes,
                f"MFCC-40, {clips} x 30 s clips @16 kHz, n_fft 2048 hop 512, 128 mels",
                "MFCC-40 audio-hours/sec")
    if name == "htdemucs":
        T = 44100 * 180
        rows = clips  # channel rows (stereo tracks * 2)
        y = 0.1 * torch.randn((rows, T), generator=gen, device=device, dtype=torch.float32)
        frames = 1 + T // 1024
        abytes = 2 * rows * (T * 4 + 2049 * frames * 8)

        def step():
            S = stft(y, n_fft=4096, hop_length=1024, pad_mode="reflect")
            return istft(S, hop_length=1024, length=T)

        return (step, rows / 2 * 180.0 / 3600.0, abytes,
                f"HTDemucs STFT+iSTFT round trip, {rows // 2} stereo x 180 s @44.1 kHz, "
                "n_fft 4096 hop 1024", "STFT round-trip track-hours/sec (n_fft 4096)")
    if name == "resample":
        T = 44100 * 30
        y = torch.randn((clips, T), generator=gen, device=device, dtype=torch.float32)
        abytes = clips * (T * 4 + 480000 * 4)
        return (lambda: resample_poly(y, 16000, 44100), clips * 30.0 / 3600.0, abytes,
                f"polyphase resampling 44.1 kHz -> 16 kHz (up 160, down 441, 56 taps per "
                f"output), {clips} x 30 s clips", "resample_poly audio-hours/sec (44.1k -> 16k)")
    if name == "resample_fft":
        T = 44100 * 30
        y = torch.randn((clips, T), generator=gen, device=device, dtype=torch.float32)
        abytes = clips * (T * 4 + 480000 * 4)
        return (lambda: resample(y, 44100, 16000), clips * 30.0 / 3600.0, abytes,
                f"Fourier-method resampling 44.1 kHz -> 16 kHz (rfft 1 323 000 -> irfft 480 000 "
                f"per clip), {clips} x 30 s clips", "resample(fft) audio-hours/sec (44.1k -> 16k)")
    if name == "features":
        from mlx_audio_amd.primitives import (spectral_bandwidth, spectral_centroid,
                                              spectral_flatness, spectral_rolloff)

        S = torch.rand((clips, 1025, 938), generator=gen, device=device, dtype=torch.float32) + 0.05
        # every statistic reads the spectrogram once (the roll-off twice) and writes one row
        abytes = 4 * clips * (1025 * 938 * 4 + 938 * 4)

        def step():
            c = spectral_centroid(S=S, sr=16000)
            return (c, spectral_bandwidth(S=S, sr=16000, centroid=c), spectral_rolloff(S=S, sr=16000),
                    spectral_flatness(S=S))

        return (step, clips * 30.0 / 3600.0, abytes,
                f"spectral centroid + bandwidth + roll-off + flatness of {clips} spectrograms "
                "(1025 bins x 938 frames = 30 s @16 kHz, n_fft 2048 hop 512)",
                "spectral features audio-hours/sec (4 statistics)")
    if name == "features_all":
        from mlx_audio_amd.primitives import spectrogram_statistics

        S = torch.rand((clips, 1025, 938), generator=gen, device=device, dtype=torch.float32) + 0.05
        abytes = clips * (1025 * 938 * 4 + 4 * 938 * 4)   # the spectrogram once, four rows out
        return (lambda: spectrogram_statistics(S, sr=16000, power=2.0), clips * 30.0 / 3600.0, abytes,
                f"centroid + bandwidth + roll-off + flatness of {clips} spectrograms (1025 x 938) in "
                "one launch", "spectral features (one launch) audio-hours/sec")
    if name == "contrast":
        from mlx_audio_amd.primitives import spectral_contrast

        S = torch.rand((clips, 1025, 938), generator=gen, device=device, dtype=torch.float32) + 0.05
        abytes = clips * (1025 * 938 * 4 + 7 * 938 * 4)
        return (lambda: spectral_contrast(S=S, sr=16000), clips * 30.0 / 3600.0, abytes,
                f"spectral contrast (6 octave bands + residual, quantile 0.02) of {clips} "
                "spectrograms (1025 bins x 938 frames)", "spectral contrast audio-hours/sec")
    if name == "onset":
        from mlx_audio_amd.primitives import onset_strength

        S = torch.rand((clips, 128, 938), generator=gen, device=device, dtype=torch.float32)
        abytes = clips * (128 * 938 * 4 + 938 * 4)
        return (lambda: onset_strength(S=S), clips * 30.0 / 3600.0, abytes,
                f"onset strength (log1p flux, mean over 128 mel bands) of {clips} mel "
                "spectrograms (128 x 938 = 30 s @16 kHz)", "onset strength audio-hours/sec")
    if name == "griffinlim":
        from mlx_audio_amd.primitives import griffinlim, magnitude

        T = 220500
        y = torch.randn((clips, T), generator=gen, device=device, dtype=torch.float32)
        S = magnitude(stft(y, n_fft=2048, hop_length=512))
        bins, frames, n_iter = S.shape[1], S.shape[2], 8
        # per iteration the state that has to stream through once even if inverse STFT, forward
        # STFT and update were one pass: spectrum in and out (8 + 8 B per bin), target magnitude
        # (4 B), previous estimate in and out (8 + 8 B); plus the final signal
        abytes = clips * (n_iter * 36 * bins * frames + 4 * T)
        return (lambda: griffinlim(S, n_iter=n_iter, init="zeros"), clips * 10.0 / 3600.0, abytes,
                f"Griffin-Lim, {n_iter} iterations, {clips} x 10 s @22.05 kHz, n_fft 2048 hop 512 "
                "(zero initial phase; per iteration: fused iSTFT, fused STFT, phase update)",
                "Griffin-Lim audio-hours/sec (8 iterations)")
    if name == "spectral_gate":
        from mlx_audio_amd.primitives import spectral_gate

        T = 220500
        y = 0.05 * torch.randn((clips, T), generator=gen, device=device, dtype=torch.float32)
        tt = torch.arange(T, device=device) / 22050.0
        y = y + torch.sin(2 * 3.141592653589793 * 440 * tt) * (tt > 2.0)
        abytes = clips * 2 * T * 4   # the waveform in, the waveform out
        return (lambda: spectral_gate(y), clips * 10.0 / 3600.0, abytes,
                f"spectral gating (stationary noise estimate, default smoothing), {clips} x 10 s "
                "@22.05 kHz, n_fft 2048 hop 512", "spectral gate audio-hours/sec")
    if name == "demucs_front":
        from mlx_audio_amd.frontends import htdemucs_normalize

        T = 44100 * 60
        mix = 0.1 * torch.randn((clips, 2, T), generator=gen, device=device, dtype=torch.float32)
        frames = -(-T // 1024)
        abytes = clips * 2 * (T * 4 + 2048 * frames * 8 + T * 4)
        return (lambda: htdemucs_normalize(mix), clips * 60.0 / 3600.0, abytes,
                f"HTDemucs model input: spectrogram (n_fft 4096, hop 1024) + complex-as-channels "
                f"normalisation + waveform normalisation, {clips} stereo x 60 s @44.1 kHz",
                "HTDemucs front-end track-hours/sec")
    if name == "features_y":
        from mlx_audio_amd.primitives import fused_statistics

        T = 480000
        y = torch.randn((clips, T), generator=gen, device=device, dtype=torch.float32)
        abytes = clips * (T * 4 + 4 * 938 * 4)
        return (lambda: fused_statistics(y, sr=16000, n_fft=2048, hop_length=512),
                clips * 30.0 / 3600.0, abytes,
                f"centroid + bandwidth + roll-off + flatness from the waveform in one launch, "
                f"{clips} x 30 s @16 kHz, n_fft 2048 hop 512", "spectral features (from audio) audio-hours/sec")
    if name in ("whisper_stem", "whisper_stem_f16"):
        from mlx_audio_amd.frontends import WhisperEncoderStem, sinusoids

        # whisper_stem: float32 weights in float32 arithmetic (the reference's default);
        # whisper_stem_f16: the float16-checkpoint arithmetic, opted into
        prec = "f16" if name.endswith("_f16") else "f32"
        n_mels, n_state, frames = 128, 1280, 3000
        mel = torch.randn((clips, n_mels, frames), generator=gen, device=device, dtype=torch.float32) * 0.5
        w1 = torch.randn((n_state, 3, n_mels), generator=gen, device=device) / (3 * n_mels) ** 0.5
        w2 = torch.randn((n_state, 3, n_state), generator=gen, device=device) / (3 * n_state) ** 0.5
        b1 = 0.1 * torch.randn((n_state,), generator=gen, device=device)
        b2 = 0.1 * torch.randn((n_state,), generator=gen, device=device)
        stem = WhisperEncoderStem(w1, b1, w2, b2, sinusoids(frames // 2, n_state), precision=prec)
        osz = 2 if prec == "f16" else 4
        abytes = clips * (n_mels * frames * 4 + (frames // 2) * n_state * osz)
        # multiply-adds of the two convolutions, 2 flops each
        WORKLOAD_FLOPS[name] = 2.0 * clips * (frames * n_state * 3 * n_mels + (frames // 2) * n_state * 3 * n_state)
        arith = ("float16 operands, float32 accumulation, float16 out" if prec == "f16" else
                 "float32 operands on the float32 matrix instruction, float32 out")
        return (lambda: stem(mel), clips * 30.0 / 3600.0, abytes,
                f"Whisper-large-v3 encoder stem (conv1 128->1280 k3 + GELU + conv2 1280->1280 k3 s2 + GELU + "
                f"positional embedding) on {clips} log-mel spectrograms (128 x 3000), {arith}",
                "Whisper encoder stem audio-hours/sec")
    raise SystemExit(f"unknown workload {name}")


WORKLOAD_FLOPS: dict = {}   # workloads bound by the matrix cores: flops per step
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X dense f16 / bf16 matrix peak
MFMA_F32_PEAK_TFLOPS = 157.3   # float32-input matrix instructions: the vector rate

KERNEL_OF = {
    "whisper_stem": "stem_gemm_kernel<float> (implicit-GEMM Conv1d on v_mfma_f32_32x32x2_f32, LDS-DMA operand "
                    "tiles, fused bias + GELU + embedding) x 2 after stem_prep_kernel",
    "whisper_stem_f16": "stem_gemm_kernel<half> (implicit-GEMM Conv1d on v_mfma_f32_32x32x16_f16, LDS-DMA "
                        "operand tiles, fused bias + GELU + embedding) x 2 after stem_prep_kernel",
    "whisper128": "logmel400d_kernel (fused pad+frame+window+rFFT-400+128-mel+log10; LDS-DMA staging, compiled mel sweep, line-aligned stores)",
    "clap64": "fwd_kernel<1024, mel> (fused pad+frame+window+rFFT+mel+dB)",
    "mfcc40": "fwd_kernel<2048, mel> (fused rFFT+mel+dB) followed by dct2_mfma_kernel",
    "htdemucs": "fwd_kernel<4096> then istft_own_kernel<4096> (fused inverse FFT + overlap-add)",
    "resample": "upfirdn_shared_kernel (polyphase FIR: input span staged in LDS, sample windows shared between adjacent outputs in registers, taps through the scalar cache)",
    "contrast": "spectral_contrast_kernel<16> (one thread per frame, sorted register arrays)",
    "demucs_front": "fwd_kernel<4096> + cac_moments_kernel + cac_normalize_kernel + row kernels",
    "spectral_gate": "fwd_kernel<2048> + magnitude + energy / threshold / mask kernels + gate_apply_spectrum_kernel + istft_own_kernel<2048>",
    "griffinlim": "per iteration istft_own_kernel<2048> + fwd_kernel<2048> + griffinlim_update_kernel",
    "onset": "onset_flux_shift_kernel (a wave per 64 frames, lagged values by lane shift)",
    "features_y": "fwd_kernel<2048> with the statistics epilogue (no spectrogram in memory)",
    "features_all": "spectral_stat_kernel, all four statistics in two passes over the bins",
    "features": "spectral_stat_kernel x 4 (one thread per frame, bins walked in order)",
    "resample_fft": "stockham_super_kernel x 4 (661 500 = 900 x 735 and 240 000 = 500 x 480 points, two "
                    "super-passes each through LDS) + resize_packed_kernel + unpack_real_kernel",
}


def measured_traffic(workload: str, clips: int):
    """(HBM bytes per step, provenance) from the committed PMC passes (FETCH_SIZE doubled as
    gfx950 needs, plus WRITE_SIZE).  The bytes are None when that workload / size has no pass
    or when a kernel source has changed since the pass (mlx_audio_amd/provenance.py)."""
    from mlx_audio_amd.provenance import measured_traffic as lookup

    return lookup(workload, clips)


DEFAULT_CLIPS = {"whisper128": 1024, "clap64": 1024, "mfcc40": 4096, "htdemucs": 128,
                 "resample": 512, "resample_fft": 64, "features": 512, "contrast": 512,
                 "onset": 4096, "griffinlim": 64, "spectral_gate": 64, "demucs_front": 16,
                 "features_y": 512, "features_all": 512, "whisper_stem": 64, "whisper_stem_f16": 64}
SECONDARY = ("clap64", "mfcc40", "htdemucs")  # BASELINE.json configs[3], [4], [2] at their per-GPU sizes


def run_secondary(device, seed: int, steps: int = 5, warmup: int = 3) -> dict:
    """The other BASELINE configurations, a few steps each, after (and outside) the headline's
    timed region; every workload's tensors are released before the next one is built."""
    import torch

    res = {}
    for name in SECONDARY:
        clips = DEFAULT_CLIPS[name]
        step, units, abytes, desc, metric = make_workload(name, clips, device, seed)
        for _ in range(warmup):
            out = step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            out = step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        gbs = abytes / (ms * 1e-3) / 1e9
        traffic, src = measured_traffic(name, clips)
        res[name] = {"workload": desc, "metric": metric, "clips_per_gpu": clips, "steps": steps,
                     "ms_per_step": ms, "value": units / (ms * 1e-3),
                     "unit": "track-hours/s" if name == "htdemucs" else "audio-hours/s",
                     "algorithmic_bytes": abytes, "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": src["status"],
                     "kernel": KERNEL_OF.get(name)}
        del step, out
        torch.cuda.empty_cache()
    return res


def cpu_baseline(budget_s: float = 8.0) -> dict:
    """The CPU oracle (NumPy float32 pocketfft + sgemm, the reference's op
    sequence) on the reference's CPU-runnable case: Whisper 80-mel log-mel of
    16 synthetic 30 s clips, one call per clip - on one thread, and with a
    ThreadPoolExecutor over the clips (NumPy's FFT and BLAS release the GIL), BLAS
    pinned to one thread per worker.  `value` is the threaded figure and `cores`
    the workers it used; the single-thread figure rides along."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    import oracle.spectral as o

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    rng = np.random.default_rng(42)
    x = rng.standard_normal((16, 480000), dtype=np.float32)
    # the box's CPU share: OMP_NUM_THREADS is set to it there (os.cpu_count() shows the host)
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1)
    workers = max(1, min(16, share))

    def one(i):
        o.compute_log_mel_spectrogram(x[i], n_mels=80)

    def timed(one_pass, budget):
        one_pass()  # warm-up
        times = []
        t_end = time.perf_counter() + budget
        while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 20):
            t0 = time.perf_counter()
            one_pass()
            times.append(time.perf_counter() - t0)
        return times

    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    if ctx is not None:
        ctx.__enter__()
    try:
        t1 = timed(lambda: [one(i) for i in range(16)], budget_s)
        with ThreadPoolExecutor(max_workers=workers) as pool:
            tn = timed(lambda: list(pool.map(one, range(16))), budget_s)
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    hours = 16 * 30.0 / 3600.0
    return {
        "value": hours / (sum(tn) / len(tn)),
        "unit": "audio-hours/s",
        "cores": workers,
        "kind": "port",
        "single_thread_value": hours / (sum(t1) / len(t1)),
        "sample": f"oracle (NumPy pocketfft f32) Whisper 80-mel log-mel, 16 x 30 s clips, one "
                  f"call per clip; {len(tn)} timed passes with {workers} worker threads over the "
                  f"clips, {len(t1)} on one thread; host shows {os.cpu_count()} logical CPUs",
    }


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this same command
    under torch.distributed.run (one process per GPU) and pass their output through.  Runs
    before anything in this process has touched the GPU."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def launcher_selftest(args, world: int, rank: int) -> None:
    """--workload none: no device work at all - forms the process group (gloo on CPUs, or
    nccl), runs the timing harness around an empty step and prints the line, so that the
    launcher path (`--gpus N` spawning N ranks) can be exercised without GPUs."""
    import torch
    import torch.distributed as dist

    from mlx_audio_amd.sharding import max_over_ranks

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
    n = dist.get_world_size() if world > 1 else 1
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (no device work)", "value": None,
                          "unit": None, "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": None, "data": "none",
                          "config": {"workload": "none", "backend": args.backend}}))
    if world > 1:
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="whisper128")
    ap.add_argument("--clips", type=int, default=None, help="clips (rows) per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short runs of the other BASELINE configurations that follow the "
                         "default workload's timed region at N = 1")
    ap.add_argument("--gather", action="store_true",
                    help="also time the run with every step's features collected on rank 0 "
                         "(chunked sends on a side stream, overlapping the next step; N > 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (gloo + --workload none: launcher self-test on CPUs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch

    from mlx_audio_amd.sharding import max_over_ranks

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.workload == "none":
        return launcher_selftest(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, device_id=device)
    # the number of ranks the process group actually formed
    n_gpus = dist.get_world_size() if dist is not None else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the process group has {n_gpus} rank(s)",
              file=sys.stderr)

    clips = args.clips or DEFAULT_CLIPS.get(args.workload, 1024)
    step, units, abytes, desc, metric = make_workload(args.workload, clips, device, 1234 + rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # The GPU comes out of idle at a low clock and needs ~20 ms of work to reach its
    # running clock (a 5-step run right after start-up measured 1.34 ms/step against 1.07
    # in steady state).  A fixed, untimed 30 ms of the same step brings it there, so that
    # the W warm-up steps asked for mean the same thing for every W; it is reported in
    # config.clock_warmup_ms.
    clock_warmup_ms = 30.0
    t_pre = time.perf_counter()
    clock_warmup_steps = 0
    while (time.perf_counter() - t_pre) * 1e3 < clock_warmup_ms:
        out = step()
        torch.cuda.synchronize()
        clock_warmup_steps += 1
    for _ in range(max(args.warmup, 1)):
        out = step()
    del out
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        out = step()
        ev[k][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    gather = None
    if args.gather and dist is not None and isinstance(out, torch.Tensor):
        # Outside the headline number: the same steps with every step's features collected
        # on rank 0 - chunked point-to-point transfers on a side stream, so the transfer of
        # step k runs beside the kernels of step k + 1 (two feature buffers alternate).
        from mlx_audio_amd.sharding import FeatureGather

        g = FeatureGather(n_gpus * clips, out.shape[1:], out.dtype, device)
        g.finish(g.start(out))  # warm-up (communicator set-up)
        barrier()
        tg = time.perf_counter()
        g.finish(g.start(out))
        barrier()
        alone_ms = max_over_ranks(time.perf_counter() - tg, device) * 1e3
        tg = time.perf_counter()
        work = None
        for k in range(args.steps):
            nxt = step()
            if work is not None:
                g.finish(work)
            work = g.start(nxt)
        g.finish(work)
        barrier()
        with_ms = max_over_ranks(time.perf_counter() - tg, device) * 1e3 / args.steps
        gather = {"gather_alone_ms": alone_ms, "ms_per_step_with_gather": with_ms,
                  "value_with_gather": n_gpus * units / (with_ms * 1e-3),
                  "note": "features of every step collected on rank 0 over xGMI, overlapped "
                          "with the next step's kernels"}
    del out
    elapsed = max_over_ranks(elapsed, device)

    secondary = None
    if (rank == 0 and n_gpus == 1 and args.workload == "whisper128" and args.clips is None
            and not args.no_secondary):
        del step
        torch.cuda.empty_cache()
        secondary = run_secondary(device, 1234)

    if rank == 0:
        value = n_gpus * units * args.steps / elapsed
        achieved = abytes / (dev_ms * 1e-3) / 1e9
        traffic, traffic_source = measured_traffic(args.workload, clips)
        line = {
            "metric": metric,
            "value": value,
            "unit": "track-hours/s" if args.workload in ("htdemucs", "demucs_front") else "audio-hours/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (random, generated on device; see config.workload)",
            "config": {"workload": desc, "clips_per_gpu": clips,
                       "parallelism": f"clips sharded over {n_gpus} GPU(s), no collective",
                       "clock_warmup_ms": clock_warmup_ms, "clock_warmup_steps": clock_warmup_steps},
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": KERNEL_OF.get(args.workload, "fwd_kernel") + ", timed with stream "
                          "events around each step (every launch of the step, e.g. the key "
                          "initialisation and the floor pass of the mel workloads, is inside "
                          "the measured time)",
                "algorithmic_bytes_per_step": abytes,
                "device_ms_per_step": dev_ms,
            },
        }
        if args.workload in WORKLOAD_FLOPS:
            tf = WORKLOAD_FLOPS[args.workload] / (dev_ms * 1e-3) / 1e12
            half = args.workload.endswith("_f16")
            peak = MFMA_F16_PEAK_TFLOPS if half else MFMA_F32_PEAK_TFLOPS
            line["dtype"] = "f16 (f32 accumulate)" if half else "f32"
            line["roofline"].update({"bound": "mfma", "achieved": tf, "peak": peak,
                                     "unit": "TFLOP/s", "frac": tf / peak,
                                     "flops_per_step": WORKLOAD_FLOPS[args.workload]})
        if gather is not None:
            line["gather"] = gather
        if secondary is not None:
            line["secondary"] = secondary
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
