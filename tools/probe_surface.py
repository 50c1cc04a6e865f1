"""A scanner-like cloud: points on a sphere surface (a 2-D manifold in 3-D, what real scans are), targets near it, K = 20 (the reference's K):
default routing against every target on the wave kernel (dev probe)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, k = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000, 2_500_000, 20
rng = np.random.default_rng(1)
def sphere(cnt, noise):
    v = rng.standard_normal((3, cnt)).astype(np.float32)
    v /= np.linalg.norm(v, axis=0, keepdims=True)
    return (0.5 + 0.45 * v + noise * rng.standard_normal((3, cnt)).astype(np.float32)).astype(np.float32)
src = sphere(n, 1e-4); tgt = sphere(m, 1e-3)
ref = None
for mode in ("default", "cpp 2", "cpp 3", "cpp 4"):
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        if "wave" in mode:
            p.set_param("tile", 0); p.set_param("wave_force", 1)
        if "cpp" in mode:
            p.set_param("refine_cells_per_point", float(mode.split()[1]))
        if "macros" in mode:
            p.set_param("refine_macros", float(mode.split()[3]))
        t0 = time.time(); p.build(src); t1 = time.time()
        p.set_targets(tgt)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        for it in range(2):
            p.rebuild(); p.query_resident_dev(k, idx, d2); torch.cuda.synchronize()
        st = p.stats()
        same = None
        if ref is None:
            ref = (idx.clone(), d2.clone())
        else:
            same = bool(torch.equal(idx, ref[0]) and torch.equal(d2, ref[1]))
        print("%-18s grid %s rho_occ %.1f refine %d build %.2f knn %.2f ms, leftover %d, wave %d, nodes %d, same=%s" %
              (mode, st["grid_dim"], st["rho_occupied"], st["n_refine"], st["ms_build"], st["ms_query"], st["n_leftover"], st["n_wave"], st["n_nodes"], same), flush=True)
