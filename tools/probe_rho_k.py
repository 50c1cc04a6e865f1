"""Dev probe: cell density sweep for a given k (100M / 10M uniform)."""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rhos = [float(v) for v in sys.argv[2:]] or [5, 6, 7, 8, 10]
n, m, seed = 100_000_000, 10_000_000, 0xC3
for rho in rhos:
    with pkg.PointsTransfer(device=0, rho=rho) as p:
        p.build_synth(n, seed); p.targets_synth(m, seed)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        for it in range(2):
            p.rebuild(); p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        st = p.stats()
        print("k", k, "rho", rho, "grid", st["grid_dim"], "build %.2f knn %.2f" % (st["ms_build"], st["ms_query"]), "left", st["n_leftover"], flush=True)
