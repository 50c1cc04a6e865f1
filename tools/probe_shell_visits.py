"""Per-target phase times of the one-wave-per-target kernel on a sphere shell (instrumented build: make visits; PT_HIP_LIB=tools/_ab/libpt_visits.so).
usage: python tools/probe_shell_visits.py n m k [name=value ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, k = int(float(sys.argv[1])), int(float(sys.argv[2])), int(sys.argv[3])
rng = np.random.default_rng(1)
def sphere(cnt, noise):
    v = rng.standard_normal((3, cnt)).astype(np.float32)
    v /= np.linalg.norm(v, axis=0, keepdims=True)
    return (0.5 + 0.45 * v + noise * rng.standard_normal((3, cnt)).astype(np.float32)).astype(np.float32)
src = sphere(n, 1e-4); tgt = sphere(m, 1e-3)
with pkg.PointsTransfer(device=0, k_hint=k) as p:
    for a in sys.argv[4:]:
        p.set_param(a.split("=")[0], float(a.split("=")[1]))
    p.build(src); p.set_targets(tgt)
    idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
    for it in range(2):
        torch.cuda.synchronize(); t = time.time(); p.query_resident_dev(k, idx, d2); torch.cuda.synchronize(); dt = time.time() - t
    st = p.stats()
    print("shell n %d m %d k %d %s: grid %s rho_occ %.1f levels %d query %.2f ms, wave targets %d, nodes %d" % (n, m, k, sys.argv[4:], st["grid_dim"], st["rho_occupied"], st["n_levels"], dt * 1e3, st["n_wave"], st["n_nodes"]))
    v = d2[:, k - 1].cpu().numpy(); du = d2[:, k - 2].cpu().numpy() / 100.0; tag = d2[:, k - 4].cpu().numpy()
    ph = [d2[:, k - 5 - j].cpu().numpy() / 100.0 for j in range(5)]
    nmg = d2[:, k - 10].cpu().numpy()
    wave = tag < 0
    nodes = np.where(wave, -tag - 1, 0)
    for name, sel in (("wave, no node", wave & (nodes == 0)), ("wave, descending", wave & (nodes > 0))):
        if not sel.any():
            continue
        print("%-18s %9d targets: records mean %.0f median %.0f p99 %.0f | wave time mean %.1f us median %.1f p99 %.1f | nodes entered mean %.1f" %
              (name, sel.sum(), v[sel].mean(), np.median(v[sel]), np.percentile(v[sel], 99), du[sel].mean(), np.median(du[sel]), np.percentile(du[sel], 99), nodes[sel].mean()))
        print("    phases us: lookup %.1f | own cell %.1f | ring-1 stream %.1f | rings>1 + flush %.1f | output + blend %.1f | merges %.1f" % (tuple(p_[sel].mean() for p_ in ph) + (nmg[sel].mean(),)))
