import sys, time
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, k, seed = 1_000_000_000, 50_000_000, 8, 0xC4
with pkg.PointsTransfer(device=0) as p:
    p.build_synth(n, seed); p.targets_synth(m, seed)
    idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
    ref = None
    for rho, tile in [(6, 1), (4, 3), (4, 2), (4.2, 1), (5, 1)]:
        p.set_param("rho", rho); p.set_param("tile", tile)
        for it in range(2):
            p.rebuild(); p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        st = p.stats()
        if ref is None: ref = (idx.clone(), d2.clone())
        same = bool(torch.equal(idx, ref[0]) and torch.equal(d2, ref[1]))
        print("rho", rho, "tile", tile, "grid", st["grid_dim"], "build %.2f tsort %.2f knn %.2f" % (st["ms_build"], st["ms_sort_targets"], st["ms_query"]), "left", st["n_leftover"], "same", same, flush=True)
