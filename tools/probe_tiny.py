"""EXPERIMENT (round 3): does the tile kernel scale with resident workgroups?  Regions of <= 2560 records (three workgroups per CU, 80
VGPRs) at rho = 2.4 against the shipped geometry (4400 records, two per CU) at rho = 4, same points and targets.
    PT_HIP_LIB=tools/_ab/libpt_tiny.so python tools/probe_tiny.py 2.4"""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
rho = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n, m, k, seed = 500_000_000, 25_000_000, 8, 0xC4
with pkg.PointsTransfer(device=0, rho=rho) as p:
    p.build_synth(n, seed); p.targets_synth(m, seed)
    idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
    rgb = torch.empty((m, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((m, 3), dtype=torch.float32, device="cuda")
    for it in range(3):
        p.query_resident_dev(k, idx, d2); torch.cuda.synchronize()
        st = p.stats()
    q = st["ms_query"]; lo = st["n_leftover"]
    for it in range(2):
        p.query_blend_resident_dev(k, 0, idx, d2, rgb, nrm); torch.cuda.synchronize()
    sb = p.stats()
    print(sys.argv[1:], "grid", st["grid_dim"], "query %.3f ms (leftover %d), with blend %.3f ms" % (q, lo, sb["ms_query"]), "checksum", int(idx.to(torch.int64).sum().item()), flush=True)
