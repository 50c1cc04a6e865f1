"""C4 (1B uniform / 50M / k=8): the query on the tile kernel (default) against every target on the one-wave-per-target kernel (dev probe)."""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, k, seed = 1_000_000_000, 50_000_000, 8, 0xC4
if len(sys.argv) > 1:
    n, m = int(float(sys.argv[1])), int(float(sys.argv[2]))
ref = None
for mode in ("tile", "wave", "wave rho 8", "wave rho 16"):
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        if mode != "tile":
            p.set_param("tile", 0); p.set_param("wave_force", 1); p.set_param("wave_min", 1)
        if "rho" in mode:
            p.set_param("rho", float(mode.split()[-1]))
        p.build_synth(n, seed); p.targets_synth(m, seed)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        for it in range(2):
            p.rebuild(); p.query_resident_dev(k, idx, d2); torch.cuda.synchronize()
        st = p.stats()
        same = None
        if ref is None:
            ref = (idx.clone(), d2.clone())
        else:
            same = bool(torch.equal(idx, ref[0]) and torch.equal(d2, ref[1]))
        print("%-12s grid %s build %.2f tsort %.2f knn %.2f ms, wave targets %d, leftover %d, same=%s" % (mode, st["grid_dim"], st["ms_build"], st["ms_sort_targets"], st["ms_query"], st["n_wave"], st["n_leftover"], same), flush=True)
