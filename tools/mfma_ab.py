# A/B of the mel projection of the n_fft = 400 kernel on one box, same binary: compiled sweep,
# table-driven sweep and the v_mfma_f32_16x16x4_f32 projection, all on the register-staging
# kernel (the one the matrix-core mode exists in), plus the shipped LDS-DMA kernel.
# "dpp" is the LDS-DMA kernel with its radix-16 pass across the lanes (DPP operands) instead of
# through LDS.
# usage: mfma_ab.py [variant]   variant in {dma, dpp, compiled, table, mfma}: run only that one
# (for counter collection), else time all four.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_audio_amd import _lib
from mlx_audio_amd._runtime import get_plan
from mlx_audio_amd.frontends import compute_log_mel_spectrogram
from mlx_audio_amd.primitives.mel import mel_filterbank_np
from mlx_audio_amd.primitives.windows import padded_window_np

lib = _lib.lib()
plan = get_plan(torch.device("cuda", 0), 400, 160, True, "constant", padded_window_np("hann", 400, 400),
                mel_basis=mel_filterbank_np(16000, 400, 128, 0.0, 8000.0, False, "slaney"), power=2.0)
OPTS = {"dma": {}, "dpp": {"w400_dpp_pass2": 1}, "compiled": {"w400_vgpr_staging": 1}, "table": {"w400_vgpr_staging": 1, "w400_table_sweep": 1},
        "mfma": {"w400_vgpr_staging": 1, "w400_mfma_mel": 1}}
y = torch.randn((1024, 480000), device="cuda")


def run(name, steps):
    for k in ("w400_vgpr_staging", "w400_table_sweep", "w400_mfma_mel", "w400_dpp_pass2"):
        lib.mia_plan_set_option(plan.handle, k.encode(), OPTS[name].get(k, 0))
    for _ in range(3):
        compute_log_mel_spectrogram(y, n_mels=128, per_clip=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        compute_log_mel_spectrogram(y, n_mels=128, per_clip=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


only = sys.argv[1] if len(sys.argv) > 1 else None
if only:
    print(only, "%.4f ms/step" % run(only, 10))
else:
    for rnd in range(3):
        print("round", rnd, " ".join("%s=%.4f" % (n, run(n, 30)) for n in OPTS))
