import sys, time
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, seed = 100_000_000, 10_000_000, 0xC3
force = len(sys.argv) > 1 and sys.argv[1] == "wave"          # tile-kernel leftovers on the one-wave-per-target kernel instead of the group kernel
for k in ((8, 20, 24, 32) if force else (1, 4, 8, 12, 16, 20, 24, 32)):
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        if force:
            p.set_param("wave_force", 1)
        p.build_synth(n, seed); p.targets_synth(m, seed)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        for it in range(2):
            p.rebuild(); p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        st = p.stats()
        print("k", k, "grid", st["grid_dim"], "build %.2f tsort %.2f knn %.2f" % (st["ms_build"], st["ms_sort_targets"], st["ms_query"]), "left", st["n_leftover"], "->", round(m / st["ms_query"] / 1e3, 1), "M targets/s (knn only)", flush=True)
