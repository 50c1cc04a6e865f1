"""k = 32 on a uniform cloud (100M / 10M): the wide tile kernel against one wave per target (dev probe)"""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, seed = 100_000_000, 10_000_000, 0xC3
ref = {}
for k, force, rho in ((32, 0, 0), (32, 1, 0), (32, 1, 12), (32, 1, 24), (32, 1, 48), (24, 1, 0), (20, 1, 0)):
    kw = dict(k_hint=k) if not rho else dict(rho=float(rho))
    with pkg.PointsTransfer(device=0, **kw) as p:
        p.set_param("wave_force", force)
        p.build_synth(n, seed); p.targets_synth(m, seed)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        for it in range(3):
            p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        st = p.stats()
        same = None
        if k in ref: same = bool(torch.equal(idx, ref[k]))
        else: ref[k] = idx.clone()
        print("k", k, "wave_force", force, "rho", rho or "hint", "grid", st["grid_dim"], "knn %.2f ms" % st["ms_query"], "wave", st["n_wave"], "->", round(m / st["ms_query"] / 1e3, 1), "M targets/s same", same, flush=True)
