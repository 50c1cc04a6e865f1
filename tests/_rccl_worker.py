"""Two-rank worker of test_native_exchange_two_ranks_over_rccl (launched by torch.distributed.run, one rank per GPU): slab build, home
search, native RCCL exchange (pt_exchange_merge_dev), result compared with the oracle over the whole cloud."""
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
from oracle import oracle as O  # noqa: E402

pkg = g.load_package()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("gloo")                                # only carries the 128-byte RCCL id and the final barrier
p = pkg.PointsTransfer(device=rank, k_hint=8)
uid = [p.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
p.comm_init(world, rank, uid[0])
n, m, k, seed = 200000, 12000, 8, 0x2A
src, tgt = O.synth_xyz(seed, 0, n), O.synth_xyz(seed, 1, m)
bounds = [-math.inf, float(np.median(src[0])), math.inf]
sel = np.nonzero((src[0] >= bounds[rank]) & (src[0] < bounds[rank + 1]))[0]
p.build(np.ascontiguousarray(src[:, sel]), gidx=sel.astype(np.uint32))
mine = np.nonzero((tgt[0] >= bounds[rank]) & (tgt[0] < bounds[rank + 1]))[0]
x = torch.from_numpy(np.ascontiguousarray(tgt[:, mine])).cuda()
i_ = torch.empty((len(mine), k), dtype=torch.int32, device="cuda"); d_ = torch.empty((len(mine), k), dtype=torch.float64, device="cuda")
p.query_dev(x, pkg.F32, len(mine), k, i_, d_)
st = p.exchange_merge_dev(x, pkg.F32, len(mine), k, 0, bounds, i_, d_)
torch.cuda.synchronize()
wi, wd = O.KdTree(src).query(tgt[:, mine], k)
assert np.array_equal(i_.cpu().numpy().view(np.uint32), wi) and np.array_equal(d_.cpu().numpy(), wd)
assert st["crossing"] > 0 and st["answered"] > 0
# the same job with the attribute table SHARDED: generated slabs in index order, each rank's own records only, the answers carrying theirs
q = pkg.PointsTransfer(device=rank, k_hint=8)
uid2 = [q.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid2, src=0)
q.comm_init(world, rank, uid2[0])
q.set_param("local_ids", 1)
q.build_synth(n, seed, slab_axis=0, slab_lo=bounds[rank], slab_hi=bounds[rank + 1])
assert q.num_source == len(sel)
q.set_targets(np.ascontiguousarray(tgt[:, mine]))
c_ = torch.empty((len(mine), 3), dtype=torch.float32, device="cuda"); n_ = torch.empty((len(mine), 3), dtype=torch.float32, device="cuda")
q.query_blend_resident_dev(k, pkg.BLEND_MEAN, i_, d_, c_, n_)
st2 = q.exchange_merge_dev(x, pkg.F32, len(mine), k, 0, bounds, i_, d_, pkg.BLEND_MEAN, c_, n_)
torch.cuda.synchronize()
assert np.array_equal(i_.cpu().numpy().view(np.uint32), wi) and np.array_equal(d_.cpu().numpy(), wd)
rc, rn = O.blend(wi, wd, O.synth_rgb(seed, n), O.synth_nrm(seed, n), mode=0)
assert np.abs(c_.cpu().numpy() - rc).max() / 255.0 <= 1e-5 and np.abs(n_.cpu().numpy() - rn).max() <= 1e-5
assert st2["bytes_sent"] > st["bytes_sent"]                       # 28 bytes per candidate instead of 12
print("rank %d ok" % rank, st, st2, flush=True)
dist.barrier()
q.comm_destroy(); q.close()
p.comm_destroy(); p.close()
dist.destroy_process_group()
