"""Tile-kernel ablation timing (dev probe): run with PT_HIP_LIB=tools/_ab/libpt_abN.so (built with -DPT_ABLATE=N).
`python tools/probe_ablate.py <label> [blend]`: `blend` runs the fused k-NN + blend launch (what bench.py runs) instead of the plain query."""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, k, seed = 1_000_000_000, 50_000_000, 8, 0xC4
with pkg.PointsTransfer(device=0, k_hint=k) as p:
    p.build_synth(n, seed); p.targets_synth(m, seed)
    idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
    blend = "blend" in sys.argv[1:]
    rgb = torch.empty((m, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((m, 3), dtype=torch.float32, device="cuda")
    for it in range(3):
        if blend:
            p.query_blend_resident_dev(k, pkg.BLEND_MEAN, idx, d2, rgb, nrm)
        else:
            p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        st = p.stats()
    print(sys.argv[1:], "knn %.3f" % st["ms_query"], "kernels", [round(v, 3) for v in st["ms_kernel"]], "leftover", st["n_leftover"], flush=True)
