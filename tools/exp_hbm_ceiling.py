"""Diagnostic: what plain streaming kernels reach on this device (the practical HBM ceiling the
LNA and merge kernels are compared with): device copy, read-only reduction, write-only fill."""
import torch
n = 1 << 30                      # 4 GiB of floats
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: b.copy_(a)); print("copy  4 GiB -> 4 GiB: %.3f ms, %.2f TB/s (read + write)" % (ms, 2 * 4 * n / ms / 1e9))
ms = t(lambda: a.sum());     print("sum   4 GiB        : %.3f ms, %.2f TB/s (read)" % (ms, 4 * n / ms / 1e9))
ms = t(lambda: b.fill_(1.0)); print("fill  4 GiB        : %.3f ms, %.2f TB/s (write)" % (ms, 4 * n / ms / 1e9))
h = a.view(torch.int32)
o = torch.empty(n, dtype=torch.int16, device="cuda")
ms = t(lambda: torch.clamp(h, -30000, 30000, out=h) if False else o.copy_(a)); print("f32 -> i16 convert : %.3f ms, %.2f TB/s (4 in + 2 out)" % (ms, 6 * n / ms / 1e9))
