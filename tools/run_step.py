"""N passes of exactly bench.py's step (rebuild + fused k-NN/blend on resident targets) for a named workload -- the command rocprofv3's
PMC passes run (`rocprofv3 --kernel-trace --pmc ... -- python3 tools/run_step.py C4 1`)."""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
cfgs = {"C2": (10_000_000, 1_000_000, 8, 0xC2), "C3": (100_000_000, 10_000_000, 16, 0xC3), "C4": (1_000_000_000, 50_000_000, 8, 0xC4),
        "C5": (1_000_000_000, 50_000_000, 32, 0xC5)}
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n, m, k, seed = cfgs[name]
with pkg.PointsTransfer(device=0, k_hint=k) as p:
    gen = dict(dist=pkg.capi.DIST_CLUSTERED, xyz_type=pkg.F16) if name == "C5" else {}
    for a in sys.argv[3:]:                      # name=value pairs for pt_set_param (probes)
        p.set_param(a.split("=")[0], float(a.split("=")[1]))
    p.build_synth(n, seed, **gen); p.targets_synth(m, seed, **gen)
    idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
    rgb = torch.empty((m, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((m, 3), dtype=torch.float32, device="cuda")
    pn = torch.empty((m, 3), dtype=torch.float32, device="cuda") if name == "C3" else None
    for it in range(steps):
        p.rebuild(); p.query_blend_resident_dev(k, pkg.BLEND_MEAN, idx, d2, rgb, nrm)
        if pn is not None:
            p.pca_normals_dev(idx, m, k, pn)          # (bench.py's C3 step ends with the PCA normals)
    torch.cuda.synchronize()
    st = p.stats(); print(name, [round(v, 3) for v in st["ms_kernel"]], "leftover", st["n_leftover"])
