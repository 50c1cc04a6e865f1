"""Random 16-byte gathers against the table size (dev probe): is the attribute gather bound by DRAM sectors or by translation?"""
import sys, time
sys.path.insert(0, '.')
import torch
dev = torch.device("cuda", 0)
G = 200_000_000                               # gathers per run
for gb in (0.125, 0.5, 2, 8, 16, 32):
    rows = int(gb * (1 << 30)) // 16
    table = torch.empty((rows, 4), dtype=torch.float32, device=dev).normal_()
    idx = torch.randint(0, rows, (G,), dtype=torch.int64, device=dev)
    out = torch.empty((G, 4), dtype=torch.float32, device=dev)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        torch.index_select(table, 0, idx, out=out)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
    # sorted indices: same gathers, DRAM/translation-friendly order
    sidx = torch.sort(idx).values
    torch.cuda.synchronize(); t0 = time.perf_counter()
    torch.index_select(table, 0, sidx, out=out)
    torch.cuda.synchronize(); ts = time.perf_counter() - t0
    print("table %6.3f GB: %.2f ms = %.1f G gathers/s random | %.2f ms = %.1f G/s sorted" % (gb, t * 1e3, G / t / 1e9, ts * 1e3, G / ts / 1e9), flush=True)
    del table, idx, out, sidx
