# Per-phase cycle stamps of the Whisper kernel; needs the library built with
# MIA_EXTRA_FLAGS="-DMIA_DEBUG_HOOKS -DMIA_W400_STAMPS" python mlx_audio_amd/csrc/build.py --force
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MIA_DBG"] = sys.argv[1] if len(sys.argv) > 1 else "8"
import torch, numpy as np
from mlx_audio_amd import _lib
from mlx_audio_amd.frontends import compute_log_mel_spectrogram
lib = _lib.lib()
lib.mia_debug_set_buffer.argtypes = [ctypes.c_void_p]
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
for itn in range(4):
    print("iter", itn)
    for w in range(16):
        print("  wave", w, " ".join("%6d" % (d[itn, w, i] - t00) for i in range(16)))
