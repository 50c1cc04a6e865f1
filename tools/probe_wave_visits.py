"""Where the one-wave-per-target kernel spends its time on the clustered cloud (instrumented build: make visits; PT_HIP_LIB=tools/_ab/libpt_visits.so).
usage: python tools/probe_wave_visits.py n m k thr"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import __graft_entry__ as g
sys.argv = sys.argv[:1] + sys.argv[1:]
pkg = g.load_package()
n, m, k, thr = int(float(sys.argv[1])), int(float(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4])
xt = pkg.F16 if k == 32 else pkg.F32
sys.path.insert(0, 'tools')
seed = 0xC5
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
def splitmix64(x):
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        return z ^ (z >> np.uint64(31))
def target_class(m, n):
    step = np.uint64(max(n // m, 1))
    i = (np.arange(m, dtype=np.uint64) * step) % np.uint64(n)
    k4 = splitmix64(np.uint64(seed) ^ (np.uint64(4) << np.uint64(56)))
    with np.errstate(over="ignore"):
        sel = splitmix64(k4 + np.uint64(4) * i)
    t = (sel % np.uint64(100)).astype(np.int64)
    return np.where(t < 70, 0, np.where(t < 95, 1, 2))
with pkg.PointsTransfer(device=0, k_hint=k) as p:
    p.set_param("refine_threshold", thr)
    for a in sys.argv[5:]:
        p.set_param(a.split("=")[0], float(a.split("=")[1]))
    p.build_synth(n, seed, dist=pkg.capi.DIST_CLUSTERED, xyz_type=xt); p.targets_synth(m, seed, dist=pkg.capi.DIST_CLUSTERED, xyz_type=xt)
    idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
    for it in range(2):
        torch.cuda.synchronize(); t = time.time(); p.query_resident_dev(k, idx, d2); torch.cuda.synchronize(); dt = time.time() - t
    st = p.stats()
    print("n %d m %d k %d thr %d: query %.1f ms, wave targets %d, nodes %d" % (n, m, k, thr, dt * 1e3, st["n_wave"], st["n_nodes"]))
    v = d2[:, k - 1].cpu().numpy(); du = d2[:, k - 2].cpu().numpy() / 100.0; t0 = d2[:, k - 3].cpu().numpy(); tag = d2[:, k - 4].cpu().numpy()
    cls = target_class(m, n)
    ph = [d2[:, k - 5 - j].cpu().numpy() / 100.0 for j in range(5)] if k >= 12 else None
    nmg = d2[:, k - 10].cpu().numpy() if k >= 12 else None
    wave = tag < 0
    nodes = np.where(wave, -tag - 1, 0)
    for name, sel in (("wave, no node", wave & (nodes == 0)), ("wave, descending", wave & (nodes > 0))):
        if not sel.any():
            continue
        print("%-18s %9d targets: records mean %.0f median %.0f p99 %.0f | wave time mean %.1f us median %.1f p99 %.1f max %.1f | sum %.1f wave-s | nodes entered mean %.1f max %d | launch span %.1f ms" %
              (name, sel.sum(), v[sel].mean(), np.median(v[sel]), np.percentile(v[sel], 99), du[sel].mean(), np.median(du[sel]), np.percentile(du[sel], 99), du[sel].max(), du[sel].sum() / 1e6,
               nodes[sel].mean(), nodes[sel].max(), ((t0[sel] + du[sel] * 100).max() - t0[sel].min()) / 1e5))
        for c, cn in enumerate(("sheet", "blob", "background")):
            w = sel & (cls == c)
            if w.any() and ph is not None:
                print("    %-10s phases us: lookup %.1f | own cell %.1f | ring-1 stream %.1f | rings>1 + flush %.1f | output + blend %.1f | merges %.1f" %
                      ((cn,) + tuple(p_[w].mean() for p_ in ph) + (nmg[w].mean(),)))
            if w.any():
                print("    %-10s %9d targets, records mean %.0f, time mean %.1f us (%.1f %% of this kernel's wave time), us per 1000 records %.2f" %
                      (cn, w.sum(), v[w].mean(), du[w].mean(), 100 * du[w].sum() / du[sel].sum(), du[w].sum() / (v[w].sum() / 1000)))
