"""BASELINE config 4 (1B / 50M / k = 8) as G logical slabs on ONE GPU: what `bench.py --gpus G` computes, with device copies in place of RCCL
(pt_exchange_merge_local) -- every slab generated in index order with positions in its records and its own attribute records -- compared row by
row with the single-context run of the whole cloud: indices and distances bit for bit, blends within 1e-5.  Prints per-slab phase times.
usage: python tools/rehearse_slabs_c4.py [G=8] [n=1e9] [sharded=1] [C4|C5] [axis=2]   (C5: the clustered fp16 cloud at k = 32, equal-count slabs from a sample's quantiles, as bench.py cuts them)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
from pt_amd import sharding
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
sharded = int(sys.argv[3]) if len(sys.argv) > 3 else 1
work = sys.argv[4] if len(sys.argv) > 4 else "C4"
axis = int(sys.argv[5]) if len(sys.argv) > 5 else 2          # the axis the slabs are cut along (bench.py: z)
m, k, seed = n // 20, (32 if work == "C5" else 8), (0xC5 if work == "C5" else 0xC4)
gen = dict(dist=pkg.capi.DIST_CLUSTERED, xyz_type=pkg.F16) if work == "C5" else {}
dev = torch.device("cuda", 0)
bounds = sharding.uniform_slab_bounds(G)
if work == "C5":
    with pkg.PointsTransfer(device=0) as probe:
        probe.build_synth(4_000_000, seed, **gen); probe.targets_synth(1_000_000, seed, **gen)
        sx = torch.empty((3, probe.num_targets), dtype=torch.float32, device=dev); probe.resident_target_xyz_dev(sx)
        bounds = sharding.quantile_slab_bounds(sx[axis], G)
t0 = time.time()
ref = pkg.PointsTransfer(device=0, k_hint=k)
ref.build_synth(n, seed, **gen); ref.targets_synth(m, seed, **gen)
ri = torch.empty((m, k), dtype=torch.int32, device=dev); rd = torch.empty((m, k), dtype=torch.float64, device=dev)
rc = torch.empty((m, 3), dtype=torch.float32, device=dev); rn = torch.empty((m, 3), dtype=torch.float32, device=dev)
ref.query_blend_resident_dev(k, pkg.BLEND_MEAN, ri, rd, rc, rn); torch.cuda.synchronize()
ref.close()
print("single context: %.1f s, %.1f GB allocated by torch" % (time.time() - t0, torch.cuda.memory_allocated() / 1e9), flush=True)
pts, xs, ii, dd, cc, nn, ids = [], [], [], [], [], [], []
tb = tq = 0.0
for s in range(G):
    p = pkg.PointsTransfer(device=0, k_hint=k)
    p.set_param("local_ids", sharded)
    p.build_synth(n, seed, slab_axis=axis, slab_lo=bounds[s], slab_hi=bounds[s + 1], **gen)
    p.targets_synth(m, seed, slab_axis=axis, slab_lo=bounds[s], slab_hi=bounds[s + 1], **gen)
    ml = p.num_targets
    x = torch.empty((3, ml), dtype=torch.float32, device=dev); p.resident_target_xyz_dev(x)
    t = torch.empty((ml,), dtype=torch.int32, device=dev); p.resident_target_ids_dev(t)
    i_ = torch.empty((ml, k), dtype=torch.int32, device=dev); d_ = torch.empty((ml, k), dtype=torch.float64, device=dev)
    c_ = torch.empty((ml, 3), dtype=torch.float32, device=dev); n_ = torch.empty((ml, 3), dtype=torch.float32, device=dev)
    p.rebuild(); p.query_blend_resident_dev(k, pkg.BLEND_MEAN, i_, d_, c_, n_); torch.cuda.synchronize()
    st = p.stats(); tb += st["ms_build"]; tq += st["ms_query"] + st["ms_sort_targets"]
    pts.append(p); xs.append(x); ii.append(i_); dd.append(d_); cc.append(c_); nn.append(n_); ids.append(t)
print("attributes %s; %d slabs built and searched: sources %s, targets %s; mean build %.2f ms, mean target sort + search %.2f ms" %
      ("sharded" if sharded else "replicated", G, [p.num_source for p in pts], [int(x.shape[1]) for x in xs], tb / G, tq / G), flush=True)
for rnd in range(3):                                        # the third round is the steady state (every buffer allocated)
    tb = tq = 0.0
    for s, p in enumerate(pts):
        p.rebuild(); p.query_blend_resident_dev(k, pkg.BLEND_MEAN, ii[s], dd[s], cc[s], nn[s]); torch.cuda.synchronize()
        st = p.stats(); tb += st["ms_build"]; tq += st["ms_query"] + st["ms_sort_targets"]
    before = [i.clone() for i in ii]
    torch.cuda.synchronize(); t1 = time.time()
    pkg.PointsTransfer.exchange_merge_local(pts, xs, pkg.F32, k, axis, bounds, ii, dd, pkg.BLEND_MEAN, cc, nn)
    torch.cuda.synchronize(); t2 = time.time()
    changed = sum(int((b != a).any(dim=1).sum()) for a, b in zip(before, ii))
    print("round %d: per slab build %.2f ms, target sort + search %.2f ms, exchange %.2f ms (wall of all slabs in sequence / G); %d of %d rows completed by another slab" %
          (rnd, tb / G, tq / G, (t2 - t1) * 1e3 / G, changed, m), flush=True)
bad = 0
worst_c = worst_n = 0.0
for s in range(G):
    t = ids[s].long()
    bad += int((ri[t] != ii[s]).any(dim=1).sum()) + int((rd[t] != dd[s]).any(dim=1).sum())
    worst_c = max(worst_c, float((rc[t] - cc[s]).abs().max()) / 255.0); worst_n = max(worst_n, float((rn[t] - nn[s]).abs().max()))
print("rows that differ from the single-context run: %d of %d; blend max error colour %.2e normal %.2e" % (bad, m, worst_c, worst_n), flush=True)
assert bad == 0 and worst_c <= 1e-5 and worst_n <= 1e-5
print("OK")
