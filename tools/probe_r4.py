"""Round-4 probe: where the non-uniform clouds stand before the round's changes (dev probe).
  dup     C5's cloud: how many DISTINCT positions, and how the duplicates are distributed
  shell   50M-point sphere shell / 2.5M targets / K = 20 under the routing knobs that exist
  clus    clustered fp32 100M / 5M / k = 8
usage: python tools/probe_r4.py dup|shell|clus [n]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
what = sys.argv[1]
dev = torch.device("cuda", 0)

if what == "dup":
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
    with pkg.PointsTransfer(device=0, k_hint=32) as p:
        p.build_synth(n, 0xC5, xyz_type=pkg.F16, dist=pkg.capi.DIST_CLUSTERED)
        st = p.stats()
        print("grid", st["grid_dim"], "rho_occ %.1f max_cell %d nodes %d build %.1f ms" % (st["rho_occupied"], st["max_cell_points"], st["n_nodes"], st["ms_build"]), flush=True)
        xyz = torch.empty((3, n), dtype=torch.float16, device=dev)
        t = p.resident_source_xyz_dev(xyz)
        assert t == pkg.F16
    key = xyz.view(torch.int16).to(torch.int64) & 0xFFFF
    key = (key[0] << 32) | (key[1] << 16) | key[2]
    del xyz
    torch.cuda.synchronize(); t0 = time.time()
    key, _ = torch.sort(key)
    torch.cuda.synchronize(); print("sort %.2f s" % (time.time() - t0), flush=True)
    first = torch.ones(n, dtype=torch.bool, device=dev)
    first[1:] = key[1:] != key[:-1]
    pos = torch.nonzero(first).flatten()
    nu = pos.numel()
    runs = torch.diff(pos, append=torch.tensor([n], device=dev))
    print("n %d distinct %d (%.3f) max run %d" % (n, nu, nu / n, int(runs.max())), flush=True)
    for cap in (8, 16, 32, 64):
        print("  kept with runs truncated at %d: %.3f of n" % (cap, float(torch.clamp(runs, max=cap).sum()) / n))
    edges = [1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 1025, 4097, 1 << 30]
    for a, b in zip(edges[:-1], edges[1:]):
        msk = (runs >= a) & (runs < b)
        print("  runs of %5d..%-6d: %9d runs, %.3f of the points" % (a, b - 1, int(msk.sum()), float(runs[msk].sum()) / n))

elif what == "shell":
    n, m, k = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000, 2_500_000, 20
    rng = np.random.default_rng(1)
    def sphere(cnt, noise):
        v = rng.standard_normal((3, cnt)).astype(np.float32)
        v /= np.linalg.norm(v, axis=0, keepdims=True)
        return (0.5 + 0.45 * v + noise * rng.standard_normal((3, cnt)).astype(np.float32)).astype(np.float32)
    src = sphere(n, 1e-4); tgt = sphere(m, 1e-3)
    ref = None
    modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["default", "group cpp 2", "group cpp 4", "group cpp 8", "group cpp 8 macros 8192", "group cpp 16 macros 8192"]
    for mode in modes:
        w = mode.split()
        with pkg.PointsTransfer(device=0, k_hint=k) as p:
            if "group" in w:
                p.set_param("wave_min", 0)
            if "tile" in w:
                p.set_param("tile", 2)
            if "tilec" in w:
                p.set_param("tile_contrast", 1)
            if "sparse" in w:
                p.set_param("tile_sparse", float(w[w.index("sparse") + 1]))
            if "cpp" in w:
                p.set_param("refine_cells_per_point", float(w[w.index("cpp") + 1]))
            if "macros" in w:
                p.set_param("refine_macros", float(w[w.index("macros") + 1]))
            if "rho" in w:
                p.set_param("rho", float(w[w.index("rho") + 1]))
            p.build(src)
            p.set_targets(tgt)
            idx = torch.empty((m, k), dtype=torch.int32, device=dev); d2 = torch.empty((m, k), dtype=torch.float64, device=dev)
            for it in range(3):
                p.rebuild(); p.query_resident_dev(k, idx, d2); torch.cuda.synchronize()
            st = p.stats()
            same = None
            if ref is None:
                ref = (idx.clone(), d2.clone())
            else:
                same = bool(torch.equal(idx, ref[0]) and torch.equal(d2, ref[1]))
            print("%-28s grid %s rho_occ %.1f refine %d levels %d build %.2f knn %.2f ms (%.0f M targets/s), leftover %d, wave %d, same=%s" %
                  (mode, st["grid_dim"], st["rho_occupied"], st["n_refine"], st["n_levels"], st["ms_build"], st["ms_query"], m / st["ms_query"] / 1e3, st["n_leftover"], st["n_wave"], same), flush=True)

elif what in ("clus", "c5"):
    n, m, k = (int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000, 5_000_000, 8) if what == "clus" else (int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000, 50_000_000, 32)
    xt = pkg.F32 if what == "clus" else pkg.F16
    modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["default"]
    for mode in modes:
        w = mode.split()
        with pkg.PointsTransfer(device=0, k_hint=k) as p:
            if "group" in w:
                p.set_param("wave_min", 0)
            if "cpp" in w:
                p.set_param("refine_cells_per_point", float(w[w.index("cpp") + 1]))
            if "macros" in w:
                p.set_param("refine_macros", float(w[w.index("macros") + 1]))
            if "tile" in w:
                p.set_param("tile", float(w[w.index("tile") + 1]))
            if "tilec" in w:
                p.set_param("tile_contrast", 1)
            if "thr" in w:
                p.set_param("refine_threshold", float(w[w.index("thr") + 1]))
            if "rho" in w:
                p.set_param("rho", float(w[w.index("rho") + 1]))
            p.build_synth(n, 0xC5, xyz_type=xt, dist=pkg.capi.DIST_CLUSTERED)
            p.targets_synth(m if what == "clus" else n // 20, 0xC5, xyz_type=xt, dist=pkg.capi.DIST_CLUSTERED)
            mm = p.num_targets
            idx = torch.empty((mm, k), dtype=torch.int32, device=dev); d2 = torch.empty((mm, k), dtype=torch.float64, device=dev)
            rgb = torch.empty((mm, 3), dtype=torch.float32, device=dev); nrm = torch.empty((mm, 3), dtype=torch.float32, device=dev)
            ts = []
            for it in range(4):
                torch.cuda.synchronize(); t0 = time.time()
                p.rebuild(); p.query_blend_resident_dev(k, pkg.BLEND_MEAN, idx, d2, rgb, nrm); torch.cuda.synchronize()
                ts.append((time.time() - t0) * 1e3)
            st = p.stats()
            print("%-28s grid %s levels %d rho_occ %.1f refine %d build %.2f knn %.2f ms step %.2f ms, leftover %d, wave %d nodes %d maxcell %d fin %.2f" %
                  (mode, st["grid_dim"], st["n_levels"], st["rho_occupied"], st["n_refine"], st["ms_build"], st["ms_query"], min(ts), st["n_leftover"], st["n_wave"], st["n_nodes"], st["max_cell_points"], st["ms_kernel"][5]), flush=True)
