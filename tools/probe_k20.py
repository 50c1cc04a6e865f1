"""k = 20 (the reference's K) at 100M / 10M: tile kernel time (dev probe)"""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, seed = 100_000_000, 10_000_000, 0xC3
for k, rho in ((20, 0), (20, 6), (20, 7), (24, 0), (16, 0)):
    kw = dict(k_hint=k) if not rho else dict(rho=float(rho))
    with pkg.PointsTransfer(device=0, **kw) as p:
        p.build_synth(n, seed); p.targets_synth(m, seed)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        for it in range(3):
            p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        st = p.stats()
        print("k", k, "rho", rho or "hint", "knn %.2f ms" % st["ms_query"], "left", st["n_leftover"], "->", round(m / st["ms_query"] / 1e3, 1), "M targets/s", flush=True)
