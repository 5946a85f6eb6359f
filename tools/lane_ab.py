"""A/B of the plan option "fwd_lane" on the CLAP workload (1024 x 10 s @48 kHz, n_fft 1024, hop 480,
64 bands, frame-major dB rows): the four-pass kernel against the kernel in which four frames share
a wave.  Alternates the two, HIP-event timing, ms per step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mlx_audio_amd import _lib, _runtime  # noqa: E402
from mlx_audio_amd.frontends import clap_log_mel  # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator(device="cuda").manual_seed(0)
y = torch.randn((clips, 480000), device="cuda", generator=g)
ref = clap_log_mel(y)
lib = _lib.lib()
plans = [p for p in _runtime._plans.values() if p.n_fft == 1024]


def timed(opt, steps=20):
    for p in plans:
        _lib.check(lib.mia_plan_set_option(p.handle, b"fwd_lane", opt))
    for _ in range(3):
        out = clap_log_mel(y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        out = clap_log_mel(y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, out


for rnd in range(3):
    a, oa = timed(0)
    b, ob = timed(1)
    print(f"round {rnd}: four-pass {a:.4f} ms   wave-private {b:.4f} ms   max |diff| {float((oa - ob).abs().max()):.2e} dB")
for p in plans:
    lib.mia_plan_set_option(p.handle, b"fwd_lane", 0)
