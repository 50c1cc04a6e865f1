"""Does the 256 MiB Infinity Cache absorb a write -> read hand-off between two kernels?  (dev probe)

A producer kernel writes a scratch buffer of S bytes, a consumer kernel reads it back, repeated over the SAME scratch while a
large stream (src -> dst, far beyond the cache) flows through: the shape of `pass-2 scatter -> finalize` run group by group
over a reused scratch.  Reported: time per byte of the pair against the same pair over a scratch far larger than the cache."""
import sys
import time
sys.path.insert(0, '.')
import torch

dev = torch.device("cuda", 0)
GB = 1 << 30
total = 8 * GB                       # bytes streamed per experiment
src = torch.empty(total // 4, dtype=torch.float32, device=dev).normal_()
dst = torch.empty(total // 4, dtype=torch.float32, device=dev)


def run(group_bytes, scratch_bytes, reps=3):
    """stream `total` bytes src -> scratch -> dst in groups of group_bytes; the scratch window cycles inside scratch_bytes"""
    scratch = torch.empty(scratch_bytes // 4, dtype=torch.float32, device=dev)
    ge, se = group_bytes // 4, scratch_bytes // 4
    ng = total // group_bytes
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        off = 0
        for g in range(ng):
            if off + ge > se:
                off = 0
            s = scratch[off:off + ge]
            s.copy_(src[g * ge:(g + 1) * ge])          # producer: HBM read + scratch write
            dst[g * ge:(g + 1) * ge].copy_(s)          # consumer: scratch read + HBM write
            off += ge
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


print("plain copy of %d GB: " % (total // GB), end="")
torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(src); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("%.2f ms = %.2f TB/s (read+write)" % (t * 1e3, 2 * total / t / 1e12))
for group_mb in (16, 32, 64, 128, 256):
    gb = group_mb << 20
    t_small = run(gb, gb)                 # scratch reused in place: fits the Infinity Cache when small
    t_big = run(gb, 4 * GB)               # scratch window walks through 4 GB: never cached
    print("group %4d MB: reused scratch %.2f ms, walking scratch %.2f ms (4 x %d GB of HBM traffic nominal; %d launches)"
          % (group_mb, t_small * 1e3, t_big * 1e3, total // GB, 2 * (total // gb)), flush=True)
