# Per-phase cycle stamps of the Whisper kernel; needs the library built with
# MIA_EXTRA_FLAGS="-DMIA_DEBUG_HOOKS -DMIA_W400_STAMPS" python mlx_audio_amd/csrc/build.py --force
# usage: stamps.py [dbg bits, default 8] [table]   ("table": the table-driven sweep)
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MIA_DBG"] = sys.argv[1] if len(sys.argv) > 1 else "8"
import torch, numpy as np
from mlx_audio_amd import _lib
from mlx_audio_amd._runtime import get_plan
from mlx_audio_amd.frontends import compute_log_mel_spectrogram
from mlx_audio_amd.primitives.mel import mel_filterbank_np
from mlx_audio_amd.primitives.windows import padded_window_np
lib = _lib.lib()
lib.mia_debug_set_buffer.argtypes = [ctypes.c_void_p]
plan = get_plan(torch.device("cuda", 0), 400, 160, True, "constant", padded_window_np("hann", 400, 400),
                mel_basis=mel_filterbank_np(16000, 400, 128, 0.0, 8000.0, False, "slaney"), power=2.0)
lib.mia_plan_set_option(plan.handle, b"w400_table_sweep", 1 if "table" in sys.argv else 0)
lib.mia_plan_set_option(plan.handle, b"w400_vgpr_staging", 1 if "vgpr" in sys.argv else 0)
buf = torch.zeros(4 * 16 * 16, dtype=torch.int64, device="cuda")
y = torch.randn((1024, 480000), device="cuda")
for _ in range(3):
    compute_log_mel_spectrogram(y, n_mels=128, per_clip=True)
torch.cuda.synchronize()
lib.mia_debug_set_buffer(buf.data_ptr())
compute_log_mel_spectrogram(y, n_mels=128, per_clip=True)
torch.cuda.synchronize()
d = buf.cpu().numpy().reshape(4, 16, 16)
t00 = d[0, :, 0].min()
if "table" in sys.argv or "vgpr" in sys.argv:  # the register-staging kernel
    names = ["start", "fetch", "pass1", "pass2", "pwr+vm", "barrier", "sweep", "stage"]
else:  # the LDS-DMA kernel: loop top, sweep done, pass 1, pass 2, loads issued, at / past barrier
    names = ["top", "sweep", "pass1", "pass2", "dma", "P+arrive", "released", "-"]
print("stamps relative to the earliest wave's start of tile 0; columns:", " ".join(names))
for itn in range(4):
    print("iter", itn)
    for w in range(16):
        print("  wave %2d" % w, " ".join("%6d" % (d[itn, w, i] - t00) for i in range(8)))
# mean phase durations over waves and tiles 1..3
nst = 8 if names[7] != "-" else 7
dur = np.diff(d[1:4, :, :nst].astype(np.int64), axis=2)
print("mean cycles per phase (tiles 1-3):", " ".join("%s=%d" % (names[i + 1], dur[:, :, i].mean()) for i in range(nst - 1)))
for w in range(16):
    print("  wave %2d phases" % w, " ".join("%s=%d" % (names[i + 1], dur[:, w, i].mean()) for i in range(nst - 1)))
per_tile = (d[3, :, 0] - d[1, :, 0]) / 2.0
print("cycles per tile (per wave):", " ".join("%d" % v for v in per_tile))
