"""pt_stream_query on the cases ADVICE r3 named: an fp64 cloud, and k in 25..32 -- since round 3 every chunk brings bounds, and the bounded tile
kernel only existed for fp32 clouds with k <= 24, so those went through the 8-lane group kernel in every chunk (dev probe).
usage: python tools/probe_stream64.py [n] [m]   (run once per library: PT_HIP_LIB=... for the 'before')"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m = (int(float(sys.argv[1])), int(float(sys.argv[2]))) if len(sys.argv) > 2 else (100_000_000, 5_000_000)
rng = np.random.default_rng(3)
for dtype, k in ((np.float64, 8), (np.float64, 20), (np.float32, 32), (np.float32, 8)):
    xyz = rng.random((3, n)).astype(dtype)
    tgt = rng.random((3, m)).astype(dtype)
    for order in ("random", "sorted"):
        if order == "sorted":
            xyz = xyz[:, np.argsort(xyz[0], kind="stable")]
        with pkg.PointsTransfer(device=0, k_hint=k) as p:
            p.set_targets(tgt)
            ts = []
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter(); si, sd = p.stream_query(xyz, n // 4, k=k); ts.append(time.perf_counter() - t0)
        print("%s k=%d %s order: %.3f s (checksum %d)" % (np.dtype(dtype).name, k, order, min(ts), int(si.sum() % 1000003)), flush=True)
