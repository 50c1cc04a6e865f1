"""Dev probe: PCIe-inclusive rate of the host-buffer entry points (pt_build_aos / pt_query_aos), AoS Point records."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
n, m, k = 100_000_000, 5_000_000, 8
rng = np.random.default_rng(1)
def points(cnt):
    a = np.zeros(cnt, dtype=pkg.POINT_DTYPE)
    a["ver"] = rng.random((cnt, 3), dtype=np.float32).astype(np.float64)
    a["normal"] = 0.0; a["color"] = 128
    return a
src = points(n); tgt = points(m)
with pkg.PointsTransfer(device=0, k_hint=k) as p:
    for it in range(2):
        t0 = time.perf_counter(); p.build_aos(src); t1 = time.perf_counter()
        idx, d2 = p.query_aos(tgt, k); t2 = time.perf_counter()
        st = p.stats()
        print("build_aos %d pts: %.1f ms (%.1f GB/s of 80-B records; device build %.1f ms)   query_aos %d targets: %.1f ms (device %.1f ms) -> %.1f M targets/s incl. PCIe"
              % (n, (t1 - t0) * 1e3, n * 80 / (t1 - t0) / 1e9, st["ms_build"], m, (t2 - t1) * 1e3, st["ms_query"] + st["ms_sort_targets"], m / (t2 - t1) / 1e6), flush=True)
