"""100M / 10M uniform cloud, k-NN only (no blend), one row of README's table: python tools/probe_k.py [f64]"""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
xt = pkg.F64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else pkg.F32
for k in (8, 16, 20, 24, 32):
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build_synth(100_000_000, 0xC3, xyz_type=xt); p.targets_synth(10_000_000, 0xC3, xyz_type=xt)
        m = p.num_targets
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        best = 1e9
        for it in range(4):
            p.rebuild(); p.query_resident_dev(k, idx, d2); torch.cuda.synchronize()
            best = min(best, p.stats()["ms_query"])
        st = p.stats()
        print("k %2d: search %.2f ms (%.0f M targets/s), build %.2f, leftover %d, wave %d" % (k, best, m / best / 1e3, st["ms_build"], st["n_leftover"], st["n_wave"]), flush=True)
