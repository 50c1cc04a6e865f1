"""Clustered cloud: the split between the group kernel and the one-wave-per-target kernel (dev probe).
usage: python tools/probe_wave.py [c5|100m] thr:wave_min ..."""
import sys, time
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
big = len(sys.argv) > 1 and sys.argv[1] == "c5"
n, m, k, xt = (1_000_000_000, 50_000_000, 32, pkg.F16) if big else (100_000_000, 5_000_000, 8, pkg.F32)
ref = None
for spec in sys.argv[2:]:
    parts = [int(v) for v in spec.split(":")]
    thr, wmin, macros = parts[0], parts[1], (parts[2] if len(parts) > 2 else 0)
    cpp = parts[3] if len(parts) > 3 else 0
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_param("refine_threshold", thr); p.set_param("wave_min", wmin)
        if macros:
            p.set_param("refine_macros", macros)
        if cpp:
            p.set_param("refine_cells_per_point", cpp)
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
        print("macros %d grid %s " % (macros, st["grid_dim"]), end="")
        print("n %d k %d thr %d wave_min %d: build %.2f knn %.2f ms, wave targets %d, nodes %d, same=%s" %
              (n, k, thr, wmin, st["ms_build"], st["ms_query"], st["n_wave"], st["n_nodes"], same), flush=True)
