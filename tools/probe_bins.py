"""dev probe: the build's passes at C4 against the number of macro bins pass 1 scatters into (cell density `rho` moves it):
   is the pass-1 scatter bound by its 1024 open write streams?"""
import sys
sys.path.insert(0, '.')
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, seed = 1_000_000_000, 0xC4
names = ["bbox", "hist1", "scat1", "hist2", "scat2", "final", "tsort", "knn"]
with pkg.PointsTransfer(device=0) as p:
    p.build_synth(n, seed)
    for rho in (4, 8, 16, 32, 64):
        p.set_param("rho", rho)
        for it in range(2):
            p.rebuild()
        torch.cuda.synchronize()
        st = p.stats()
        ms = list(st["ms_kernel"])
        print("rho", rho, "grid", st["grid_dim"], "macros", [(d + 63) // 64 for d in st["grid_dim"]], "build %.2f" % st["ms_build"],
              " ".join("%s %.2f" % (a, b) for a, b in zip(names, ms)), flush=True)
