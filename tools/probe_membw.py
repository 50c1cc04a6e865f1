"""Calibration (dev probe): what a plain device copy / read / fill achieves on this GPU, for the roofline discussion."""
import torch
n = 4_000_000_000  # float32 elements = 16 GB
x = torch.empty(n, dtype=torch.float32, device="cuda"); x.fill_(1.0)
y = torch.empty_like(x)
def timed(f, reps=3):
    best = 1e9
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
t = timed(lambda: y.copy_(x)); print("copy 16GB->16GB: %.2f ms  %.2f TB/s (read+write)" % (t, 32e9 / t / 1e9))
t = timed(lambda: y.fill_(2.0)); print("fill 16GB: %.2f ms  %.2f TB/s" % (t, 16e9 / t / 1e9))
t = timed(lambda: torch.sum(x)); print("sum 16GB: %.2f ms  %.2f TB/s" % (t, 16e9 / t / 1e9))
