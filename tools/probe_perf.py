"""Ad-hoc perf probe (not part of the product): per-kernel device times of one build+query at a given size."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
cfgs = {"C2": (10_000_000, 1_000_000, 8, 0xC2), "C3": (100_000_000, 10_000_000, 16, 0xC3), "C4": (1_000_000_000, 50_000_000, 8, 0xC4)}
names = sys.argv[1:] or ["C2", "C3", "C4"]
for name in names:
    n, m, k, seed = cfgs[name]
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        t = time.time(); p.build_synth(n, seed); p.targets_synth(m, seed); print(name, "setup", round(time.time() - t, 3), flush=True)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        rgb = torch.empty((m, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((m, 3), dtype=torch.float32, device="cuda")
        for it in range(3):
            torch.cuda.synchronize(); t = time.time()
            p.rebuild(); p.query_resident_dev(k, idx, d2); p.blend_dev(idx, d2, m, k, 0, rgb, nrm)
            torch.cuda.synchronize(); dt = time.time() - t
            st = p.stats()
            print(name, "wall %.2f ms" % (dt * 1e3), "build %.3f tsort %.3f knn %.3f blend %.3f" % (st["ms_build"], st["ms_sort_targets"], st["ms_query"], st["ms_blend"]),
                  "kernels", [round(v, 3) for v in st["ms_kernel"]], "leftover", st["n_leftover"], "grid", st["grid_dim"], "GB", round(st["device_bytes"] / 1e9, 2), flush=True)
