"""Grid chosen for the clustered cloud (dev probe): python tools/probe_grid.py [c5|100m]"""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
big = len(sys.argv) > 1 and sys.argv[1] == "c5"
n, m, k, xt = (1_000_000_000, 50_000_000, 32, pkg.F16) if big else (100_000_000, 5_000_000, 8, pkg.F32)
with pkg.PointsTransfer(device=0, k_hint=k) as p:
    p.build_synth(n, 0xC5, dist=pkg.capi.DIST_CLUSTERED, xyz_type=xt)
    st = p.stats()
    print({kk: st[kk] for kk in ("grid_dim", "cell_size", "n_cells", "n_levels", "rho_occupied", "n_refine", "max_cell_points", "n_nodes", "ms_build", "device_bytes")})
    print("kernels", [round(v, 2) for v in st["ms_kernel"]])
