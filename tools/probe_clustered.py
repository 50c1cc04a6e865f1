import sys, time
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, k = 100_000_000, 5_000_000, 8
for adaptive in (0, 1):
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_param("adaptive", adaptive)
        p.build_synth(n, 0xC5, dist=pkg.capi.DIST_CLUSTERED); p.targets_synth(m, 0xC5, dist=pkg.capi.DIST_CLUSTERED)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        t = time.time(); p.rebuild(); p.query_resident_dev(k, idx, d2); torch.cuda.synchronize(); dt = time.time() - t
        st = p.stats()
        print("adaptive", adaptive, "wall %.1f ms" % (dt * 1e3), "build %.2f knn %.2f" % (st["ms_build"], st["ms_query"]), "leftover", st["n_leftover"], "grid", st["grid_dim"], "rho_occ %.1f refine %d" % (st["rho_occupied"], st["n_refine"]), flush=True)
