"""Error of resample(res_type='fft') against scipy.signal.resample (float32 and float64)."""
import sys, time
import numpy as np, torch, scipy.signal as ss
sys.path.insert(0, ".")
from mlx_audio_amd.primitives import resample
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
cases = [(1, 3), (2, 3), (3, 2), (2, 1), (7, 5), (16, 8), (8, 16), (17, 16), (441, 160), (160, 441), (1000, 999),
         (1001, 1000), (4410, 1600), (44100, 16000), (100003, 36283), (65536, 32768), (32768, 65536),
         (1323000, 480000), (480000, 1323000), (1234567, 447915)]
for n, m in cases:
    x = rng.standard_normal((2, n)).astype(np.float32)
    want32 = ss.resample(x, m, axis=-1).astype(np.float32)
    want64 = ss.resample(x.astype(np.float64), m, axis=-1)
    got = resample(torch.from_numpy(x).to(dev), n, m, res_type="fft")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = resample(torch.from_numpy(x).to(dev), n, m, res_type="fft")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    got = got.cpu().numpy()
    assert got.shape == want32.shape, (got.shape, want32.shape)
    peak = np.abs(want64).max() + 1e-30
    print(f"{n:>8} -> {m:>8}: gpu-vs-f64 {np.abs(got - want64).max() / peak:.2e}  scipy32-vs-f64 "
          f"{np.abs(want32 - want64).max() / peak:.2e}  gpu-vs-scipy32 {np.abs(got - want32).max() / peak:.2e}  {dt*1e3:.2f} ms", flush=True)
