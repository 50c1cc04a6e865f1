"""Clustered cloud at 100M (fp16, k = 32) and 30M: tile kernel + leftovers against every target on the wave kernel (dev probe)."""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
for n, m, k, xt in ((100_000_000, 5_000_000, 32, pkg.F16), (30_000_000, 1_500_000, 32, pkg.F16), (30_000_000, 1_500_000, 8, pkg.F32), (100_000_000, 5_000_000, 20, pkg.F32)):
    ref = None
    for tile in (1, 0):
        with pkg.PointsTransfer(device=0, k_hint=k) as p:
            p.set_param("tile", tile)
            if not tile:
                p.set_param("wave_force", 1)
            p.build_synth(n, 0xC5, dist=pkg.capi.DIST_CLUSTERED, xyz_type=xt); p.targets_synth(m, 0xC5, dist=pkg.capi.DIST_CLUSTERED, xyz_type=xt)
            idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
            for it in range(2):
                p.rebuild(); p.query_resident_dev(k, idx, d2); torch.cuda.synchronize()
            st = p.stats()
            same = None
            if ref is None:
                ref = (idx.clone(), d2.clone())
            else:
                same = bool(torch.equal(idx, ref[0]) and torch.equal(d2, ref[1]))
            print("n %d k %d tile %d: grid %s rho_occ %.1f (rho target %.1f) build %.2f knn %.2f ms, leftover %d, wave %d, same=%s" %
                  (n, k, tile, st["grid_dim"], st["rho_occupied"], n / st["n_cells"], st["ms_build"], st["ms_query"], st["n_leftover"], st["n_wave"], same), flush=True)
