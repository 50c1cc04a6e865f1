"""One warm-up + N timed passes of the hot path for a named workload (used under rocprofv3)."""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
cfgs = {"C2": (10_000_000, 1_000_000, 8, 0xC2), "C3": (100_000_000, 10_000_000, 16, 0xC3), "C4": (1_000_000_000, 50_000_000, 8, 0xC4)}
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n, m, k, seed = cfgs[name]
with pkg.PointsTransfer(device=0) as p:
    p.build_synth(n, seed); p.targets_synth(m, seed)
    idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
    rgb = torch.empty((m, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((m, 3), dtype=torch.float32, device="cuda")
    for it in range(steps):
        p.rebuild(); p.query_resident_dev(k, idx, d2); p.blend_dev(idx, d2, m, k, 0, rgb, nrm)
    torch.cuda.synchronize()
    st = p.stats(); print(name, st["ms_kernel"], "leftover", st["n_leftover"])
