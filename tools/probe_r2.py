"""Round-2 probe at C4 (dev probe): per-kernel build times + fused query time, results of repeated runs compared."""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, k, seed = 1_000_000_000, 50_000_000, 8, 0xC4
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
with pkg.PointsTransfer(device=0, k_hint=k) as p:
    if len(sys.argv) > 2 and sys.argv[2] == "wave":
        p.set_param("wave_force", 1)                # tile-kernel leftovers on the wave kernel
    p.build_synth(n, seed); p.targets_synth(m, seed)
    idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
    rgb = torch.empty((m, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((m, 3), dtype=torch.float32, device="cuda")
    ref = None
    for r in range(reps):
        for it in range(2):
            p.rebuild()
            st = p.stats()
        for it in range(2):
            p.query_blend_resident_dev(k, 0, idx, d2, rgb, nrm); torch.cuda.synchronize()
        sq = p.stats()
        p.query_resident_dev(k, idx, d2); torch.cuda.synchronize()
        sn = p.stats()
        same = None
        if ref is None:
            ref = (idx.clone(), d2.clone(), rgb.clone(), nrm.clone())
        else:
            same = (bool(torch.equal(idx, ref[0]) and torch.equal(d2, ref[1])), float((rgb - ref[2]).abs().max()), float((nrm - ref[3]).abs().max()))
        print("build %.2f ms kernels %s | query+blend %.2f, query only %.2f (+sort %.2f) leftover %d same=%s" %
              (st["ms_build"], [round(v, 2) for v in st["ms_kernel"][:6]], sq["ms_query"], sn["ms_query"], sq["ms_sort_targets"], sq["n_leftover"], same), flush=True)
