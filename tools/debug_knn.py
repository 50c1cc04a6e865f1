import sys
sys.path.insert(0, '.')
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
from oracle import oracle as O
src, tgt = O.synth_xyz(0xC1, 0, 10000), O.synth_xyz(0xC1, 1, 1000)
with pkg.PointsTransfer(device=0) as p:
    p.build(src)
    for k in (1, 2, 7, 8, 9, 16, 20, 32):
        idx, d2 = p.query(tgt, k)
        wi, wd = O.knn_bruteforce(src, tgt, k)
        bad = np.nonzero((idx != wi).any(axis=1))[0]
        print("k", k, "bad rows", len(bad), "of", len(wi))
        for t in bad[:3]:
            print("  row", t, "got", idx[t][:10], d2[t][:4], "want", wi[t][:10], wd[t][:4])
