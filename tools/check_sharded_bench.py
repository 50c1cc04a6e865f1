"""Rehearsal of the multi-rank flow on ONE GPU: run the sharded pipeline at world size W over gloo (host-staged
collectives) and compare every rank's final neighbour lists with a single-context run of the whole cloud."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
from pt_amd import sharding
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
comm = sharding.HostStagedComm()
n, m, k, seed = 3_000_000, 300_000, 8, 0xC4
bounds = sharding.uniform_slab_bounds(world)
pt = pkg.PointsTransfer(device=0)
pt.build_synth(n, seed, slab_axis=0, slab_lo=bounds[rank], slab_hi=bounds[rank + 1])
pt.targets_synth(m, seed, slab_axis=0, slab_lo=bounds[rank], slab_hi=bounds[rank + 1])
ml = pt.num_targets
idx = torch.empty((ml, k), dtype=torch.int32, device=dev); d2 = torch.empty((ml, k), dtype=torch.float64, device=dev)
xyz = torch.empty((3, ml), dtype=torch.float32, device=dev); pt.resident_target_xyz_dev(xyz)
ids = torch.empty((ml,), dtype=torch.int32, device=dev); pt.resident_target_ids_dev(ids)
pt.query_resident_dev(k, idx, d2)
st = sharding.exchange_and_merge(comm, sharding.GpuSlabEngine(pt, pkg.F32, dev), xyz, idx, d2, k, 0, bounds)
rgb = torch.empty((ml, 3), dtype=torch.float32, device=dev); nrm = torch.empty((ml, 3), dtype=torch.float32, device=dev)
pt.blend_dev(idx, d2, ml, k, 0, rgb, nrm)
torch.cuda.synchronize()
# reference: the whole cloud in one context
ref = pkg.PointsTransfer(device=0)
ref.build_synth(n, seed); ref.targets_synth(m, seed)
ri = torch.empty((m, k), dtype=torch.int32, device=dev); rd = torch.empty((m, k), dtype=torch.float64, device=dev)
ref.query_resident_dev(k, ri, rd)
rrgb = torch.empty((m, 3), dtype=torch.float32, device=dev); rnrm = torch.empty((m, 3), dtype=torch.float32, device=dev)
ref.blend_dev(ri, rd, m, k, 0, rrgb, rnrm)
torch.cuda.synchronize()
sel = ids.long()
ok = bool(torch.equal(idx, ri[sel]) and torch.equal(d2, rd[sel]) and torch.equal(rgb, rrgb[sel]) and torch.equal(nrm, rnrm[sel]))
if not ok:
    bad = torch.nonzero((idx != ri[sel]).any(dim=1)).flatten()
    print("rank", rank, "bad rows", bad.numel(), "of", ml, flush=True)
    for b_ in bad[:3].tolist():
        print("  x=%.6f got" % float(xyz[0, b_]), idx[b_].tolist(), [float(v) for v in d2[b_][:3]], "want", ri[sel[b_]].tolist(), [float(v) for v in rd[sel[b_]][:3]], flush=True)
res = [None] * world
dist.all_gather_object(res, (ok, ml, st["crossing"], st["answered"]))
if rank == 0:
    print("SHARDED", world, res, flush=True)
    assert all(r[0] for r in res) and sum(r[1] for r in res) == m
dist.destroy_process_group()
sys.exit(0 if ok else 1)
