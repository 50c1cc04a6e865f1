"""The sphere shell (50M / 2.5M / K = 20) as the CLI would hold it: an fp64 cloud (dev probe).  usage: python tools/probe_shell64.py [n]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
m, k = n // 20, 20
rng = np.random.default_rng(1)
def sphere(cnt, noise):
    v = rng.standard_normal((3, cnt)); v /= np.linalg.norm(v, axis=0, keepdims=True)
    return 0.5 + 0.45 * v + noise * rng.standard_normal((3, cnt))
src = sphere(n, 1e-4); tgt = sphere(m, 1e-3)
for dt in (np.float32, np.float64):
    s_ = np.ascontiguousarray(src.astype(dt)); t_ = np.ascontiguousarray(tgt.astype(dt))
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build(s_); p.set_targets(t_)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        for it in range(3):
            p.rebuild(); p.query_resident_dev(k, idx, d2); torch.cuda.synchronize()
        st = p.stats()
        print("%s: grid %s rho_occ %.1f build %.2f knn %.2f ms (%.0f M targets/s) wave %d leftover %d" % (dt.__name__, st["grid_dim"], st["rho_occupied"], st["ms_build"], st["ms_query"], m / st["ms_query"] / 1e3, st["n_wave"], st["n_leftover"]), flush=True)
