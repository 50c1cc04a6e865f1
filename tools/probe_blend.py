"""dev probe: the stand-alone blend (pt_blend_dev) at 100M / 10M, k = 8 and 20"""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, seed = 100_000_000, 10_000_000, 0xC3
for k in (8, 20):
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build_synth(n, seed); p.targets_synth(m, seed)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        rgb = torch.empty((m, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((m, 3), dtype=torch.float32, device="cuda")
        p.query_resident_dev(k, idx, d2)
        for it in range(3): p.blend_dev(idx, d2, m, k, 1, rgb, nrm)
        torch.cuda.synchronize()
        print("k", k, "blend %.2f ms" % p.stats()["ms_blend"], "checksum", float(rgb.double().sum()), float(nrm.double().sum()), flush=True)
