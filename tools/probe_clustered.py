"""Clustered cloud (BASELINE config 5's distribution) at 100M / 5M: build + k-NN with and without the refined cells (dev probe)."""
import sys, time
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
cases = [(100_000_000, 5_000_000, 8, pkg.F32), (100_000_000, 5_000_000, 32, pkg.F16)]
if len(sys.argv) > 1 and sys.argv[1] == "c5":
    cases = [(1_000_000_000, 50_000_000, 32, pkg.F16)]
for n, m, k, xt in cases:
    ref = None
    for thr in ((0, 512, 2048, 8192) if n <= 100_000_000 else (0, 2048, 512)):
        with pkg.PointsTransfer(device=0, k_hint=k) as p:
            p.set_param("refine_threshold", thr)
            p.build_synth(n, 0xC5, dist=pkg.capi.DIST_CLUSTERED, xyz_type=xt); p.targets_synth(m, 0xC5, dist=pkg.capi.DIST_CLUSTERED, xyz_type=xt)
            idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
            for it in range(2):
                t = time.time(); p.rebuild(); p.query_resident_dev(k, idx, d2); torch.cuda.synchronize(); dt = time.time() - t
            st = p.stats()
            same = None
            if ref is None:
                ref = (idx.clone(), d2.clone())
            else:
                same = bool(torch.equal(idx, ref[0]) and torch.equal(d2, ref[1]))
            print("n %d k %d thr %d: wall %.1f ms, build %.2f knn %.2f, leftover %d, grid %s, rho_occ %.1f, max cell %d, nodes %d levels %d, same=%s" %
                  (n, k, thr, dt * 1e3, st["ms_build"], st["ms_query"], st["n_leftover"], st["grid_dim"], st["rho_occupied"], st["max_cell_points"],
                   st["n_nodes"], st["refine_levels"], same), flush=True)
