"""How many records does the group kernel look at per target on the clustered cloud (config 5's distribution)?  Dev probe; needs
`make -C 3d-reconstruction-from-point-cloud_amd/csrc visits` and PT_HIP_LIB=tools/_ab/libpt_visits.so (the instrumented build writes
the count into the last d2 column).  Targets are classified like the generator does (sheet / blob / background)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, k = int(float(sys.argv[1])), int(float(sys.argv[2])), int(sys.argv[3])
thr = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
xt = pkg.F16 if k == 32 else pkg.F32
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
    return np.where(t < 70, 0, np.where(t < 95, 1, 2)), ((sel >> np.uint64(8)) % np.uint64(256)).astype(np.int64)


with pkg.PointsTransfer(device=0, k_hint=k) as p:
    p.set_param("refine_threshold", thr)
    p.build_synth(n, seed, dist=pkg.capi.DIST_CLUSTERED, xyz_type=xt); p.targets_synth(m, seed, dist=pkg.capi.DIST_CLUSTERED, xyz_type=xt)
    idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
    for it in range(2):
        torch.cuda.synchronize(); t = time.time(); p.query_resident_dev(k, idx, d2); torch.cuda.synchronize(); dt = time.time() - t
    st = p.stats()
    v = d2[:, k - 1].cpu().numpy()
    print("n %d m %d k %d thr %d: query %.1f ms, leftover %d, nodes %d, max cell %d" % (n, m, k, thr, dt * 1e3, st["n_leftover"], st["n_nodes"], st["max_cell_points"]))
    cls, sub = target_class(m, n)
    tot = v.sum()
    print("records looked at: total %.3e, mean %.0f per target, median %.0f, p90 %.0f, p99 %.0f, max %.0f" %
          (tot, v.mean(), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
    for c, name in enumerate(("sheet", "blob", "background")):
        w = v[cls == c]
        print("  %-10s %9d targets, %5.1f %% of the records, mean %.0f, median %.0f, p99 %.0f, max %.0f" %
              (name, w.size, 100.0 * w.sum() / tot, w.mean(), np.median(w), np.percentile(w, 99), w.max()))
    order = np.argsort(v)[::-1]
    cum = np.cumsum(v[order]) / tot
    for frac in (0.001, 0.01, 0.1):
        print("  the heaviest %.1f %% of the targets account for %.1f %% of the records" % (100 * frac, 100 * cum[int(frac * m) - 1]))
    # per blob / per sheet totals: which primitives are expensive
    for c, name, cnt in ((0, "sheet", 64), (1, "blob", 256)):
        sel = cls == c
        per = np.bincount(sub[sel] % cnt, weights=v[sel], minlength=cnt); num = np.bincount(sub[sel] % cnt, minlength=cnt)
        top = np.argsort(per)[::-1][:6]
        print("  heaviest %ss: %s" % (name, ", ".join("#%d: %.1f %% (mean %.0f)" % (i, 100 * per[i] / tot, per[i] / max(num[i], 1)) for i in top)))
    # time line of the launch (instrumented build): per wave its start and duration on the 100 MHz wall clock
    if k >= 4:
        t0 = d2[:, k - 3].cpu().numpy(); du = d2[:, k - 2].cpu().numpy(); wv = d2[:, k - 4].cpu().numpy().astype(np.int64)
        first = np.unique(wv, return_index=True)[1]                       # one entry per wave (its 8 targets carry the same stamps... per group: take one)
        t0w, duw, vw = t0[first], du[first], np.bincount(np.unique(wv, return_inverse=True)[1], weights=v)
        base = t0w.min(); end = (t0w + duw).max()
        print("waves %d, launch %.2f ms; wave duration: mean %.1f us, median %.1f, p99 %.1f, max %.1f; sum of durations %.1f wave-ms (= %.2f ms on 4096 slots)" %
              (t0w.size, (end - base) / 1e5, duw.mean() / 100, np.median(duw) / 100, np.percentile(duw, 99) / 100, duw.max() / 100, duw.sum() / 1e5, duw.sum() / 1e5 / 4096))
        edges = np.linspace(base, end, 21)
        act = [(np.minimum(t0w + duw, edges[i + 1]) - np.maximum(t0w, edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) for i in range(20)]
        print("  waves resident over the launch (20 slices):", " ".join("%d" % a for a in act))
        heavy = np.argsort(duw)[::-1][:5]
        for h in heavy:
            tt = np.nonzero(wv == wv[first][h])[0]
            print("  slow wave: start +%.2f ms, %.1f us, records %s, classes %s" % ((t0w[h] - base) / 1e5, duw[h] / 100, v[tt].astype(np.int64).tolist(), cls[tt].tolist()))
        c = np.corrcoef(duw, vw)[0, 1]
        print("  correlation(duration, records per wave) = %.2f; us per 1000 records: %.2f" % (c, duw.sum() / 100 / (v.sum() / 1000)))
