"""Random 16-byte attribute gathers through the library's own blend kernel against the TABLE SIZE (dev probe, round 3): is the blend
epilogue's gather rate (42 G gathers/s at C4) set by DRAM sectors or by address translation?  m x k random indices into tables of
n = 1e6 .. 1e9 records (16 MB .. 16 GB)."""
import sys, time
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
m, k = 50_000_000, 8
dev = "cuda"
d2 = torch.ones((m, k), dtype=torch.float64, device=dev)
rgb = torch.empty((m, 3), dtype=torch.float32, device=dev); nrm = torch.empty((m, 3), dtype=torch.float32, device=dev)
for n in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else (1_000_000, 16_000_000, 64_000_000, 250_000_000, 1_000_000_000):
    with pkg.PointsTransfer(device=0) as p:
        p.build_synth(n, 0xC4)
        idx = torch.randint(0, n, (m, k), dtype=torch.int64, device=dev).to(torch.int32)
        for order in ("random", "sorted rows"):
            if order != "random":
                idx = torch.sort(idx.view(-1)).values.view(m, k).contiguous()
            ts = []
            for it in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                p.blend_dev(idx, d2, m, k, 0, rgb, nrm)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            t = min(ts)
            print("table %7.0f MB (%10d records), %s indices: %.2f ms = %.1f G gathers/s = %.2f TB/s of 64-byte sectors" % (n * 16 / 1e6, n, order, t * 1e3, m * k / t / 1e9, m * k * 64 / t / 1e12), flush=True)
