import torch, time
y = torch.randn((1024, 480000), device="cuda")
out = torch.empty((1024, 128, 3001), device="cuda")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: y.sum()); print("read 1.97GB (sum): %.3f ms  %.2f TB/s" % (ms, y.numel()*4/ms/1e9))
ms = t(lambda: out.fill_(1.0)); print("write 1.57GB (fill): %.3f ms  %.2f TB/s" % (ms, out.numel()*4/ms/1e9))
z = torch.empty_like(y)
ms = t(lambda: z.copy_(y)); print("copy 1.97GB: %.3f ms  %.2f TB/s (r+w)" % (ms, 2*y.numel()*4/ms/1e9))
