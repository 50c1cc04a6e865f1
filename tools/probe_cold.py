"""Cold builds of non-uniform clouds: sorts, refinements and build time of the FIRST build with and without the pre-sort refinement (dev probe).
usage: python tools/probe_cold.py shell|clus|c5|ordered|uniform [n]     (PT_DEBUG_PRESORT=1: the library prints what the sample said -- chi-square, degrees of freedom,
consecutive points sharing a block, the occupancy estimate and the grid it was taken on -- to stderr, once per probe)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
what = sys.argv[1]
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else {"shell": 50_000_000, "clus": 100_000_000, "c5": 1_000_000_000, "ordered": 64_000_000, "uniform": 64_000_000}[what]
rng = np.random.default_rng(1)
src = None
if what == "shell":
    v = rng.standard_normal((3, n)).astype(np.float32); v /= np.linalg.norm(v, axis=0, keepdims=True)
    src = (0.5 + 0.45 * v + 1e-4 * rng.standard_normal((3, n)).astype(np.float32)).astype(np.float32)
elif what in ("ordered", "uniform"):
    src = rng.random((3, n), dtype=np.float32)
    if what == "ordered":
        src = np.ascontiguousarray(src[:, np.argsort((src[0] * 64).astype(np.int32) * 4096 + (src[1] * 64).astype(np.int32) * 64 + (src[2] * 64).astype(np.int32), kind="stable")])
k = 32 if what == "c5" else (20 if what == "shell" else 8)
m = n // 20
for pre in (1, 0, 1, 0):
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_param("presort_refine", pre)
        torch.cuda.synchronize(); t0 = time.time()
        if src is not None:
            p.build(src)
        else:
            p.build_synth(n, 0xC5, xyz_type=pkg.F16 if what == "c5" else pkg.F32, dist=pkg.capi.DIST_CLUSTERED)
        torch.cuda.synchronize(); t1 = time.time()
        st = p.stats()
        if src is not None:
            p.set_targets(np.ascontiguousarray(src[:, :m] + np.float32(1e-3)))
        else:
            p.targets_synth(m, 0xC5, xyz_type=pkg.F16 if what == "c5" else pkg.F32, dist=pkg.capi.DIST_CLUSTERED)
        mm = p.num_targets
        idx = torch.empty((mm, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((mm, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, idx, d2); torch.cuda.synchronize()
        q = p.stats()["ms_query"]
        p.set_param("forget", 1); p.rebuild(); torch.cuda.synchronize()
        s2 = p.stats()
        print("   flags first:", {f: st[f] for f in ("pass1_pooled", "pass2_pooled", "bbox_guess")}, "rebuild:", {f: s2[f] for f in ("pass1_pooled", "pass2_pooled", "bbox_guess", "n_refine")})
        print("%s presort=%d: first build wall %.1f ms (ms_build %.2f), sorts %d, refine %d (presort %d), grid %s rho_occ %.1f probe %d ordered %d | query %.2f ms | forget+rebuild %.2f ms sorts %d presort %d grid %s" %
              (what, pre, (t1 - t0) * 1e3, st["ms_build"], st["n_sorts"], st["n_refine"], st["presort_refine"], st["grid_dim"], st["rho_occupied"], st["uniform_probe"], st["ordered_input"], q,
               s2["ms_build"], s2["n_sorts"], s2["presort_refine"], s2["grid_dim"]), flush=True)
