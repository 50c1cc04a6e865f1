"""One build + one query of the clustered cloud (for PMC passes): python tools/run_clustered.py n m k thr"""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, k, thr = int(float(sys.argv[1])), int(float(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4])
xt = pkg.F16 if k == 32 else pkg.F32
with pkg.PointsTransfer(device=0, k_hint=k) as p:
    p.set_param("refine_threshold", thr)
    p.build_synth(n, 0xC5, dist=pkg.capi.DIST_CLUSTERED, xyz_type=xt); p.targets_synth(m, 0xC5, dist=pkg.capi.DIST_CLUSTERED, xyz_type=xt)
    idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
    p.query_resident_dev(k, idx, d2); torch.cuda.synchronize()
    print(p.stats()["ms_query"])
