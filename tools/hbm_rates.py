"""What the box moves with plain torch kernels: fill (write only), copy (read + write), sum (read only),
on buffers far larger than the 256 MiB Infinity Cache.  TB/s of bytes moved."""
import torch
n = 4 * (1 << 30)   # 4 Gi floats = 16 GiB
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n // 4, dtype=torch.float32, device="cuda")


def timed(fn, byts, name, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:40s} {ms:8.3f} ms  {byts / ms / 1e9:6.2f} TB/s")


timed(lambda: a.fill_(1.0), 4 * n, "fill 16 GiB (write only)")
timed(lambda: a.zero_(), 4 * n, "zero 16 GiB (memset)")
timed(lambda: b.copy_(a[: n // 4]), 2 * n, "copy 4 GiB (read + write)")
timed(lambda: a[: n // 2].sum(), 2 * n, "sum 8 GiB (read only)")
a2 = a.view(2, n // 2)
timed(lambda: a2[1].copy_(a2[0]), 4 * n, "copy 8 GiB (read + write)")
