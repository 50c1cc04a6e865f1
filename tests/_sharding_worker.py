"""Worker of tests/test_sharding_gloo.py: one rank of the slab protocol on the CPU (gloo), with an
oracle-backed engine standing in for the GPU kernels.  The PROTOCOL code under test is the product's
(pt_amd.sharding.exchange_and_merge); only the three compute steps are the oracle's."""
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

NOIDX = 0xFFFFFFFF


class OracleSlabEngine:
    def __init__(self, slab_xyz, slab_gidx):
        self.src, self.gidx = slab_xyz, slab_gidx

    def slab_need(self, xyz, d2, k, axis, bounds, my_slab):
        c = xyz[axis].double().numpy()
        kth = d2[:, k - 1].numpy()
        G = len(bounds) - 1
        need = np.zeros((G, c.shape[0]), np.uint8)
        for s in range(G):
            if s == my_slab:
                continue
            lo, hi = bounds[s], bounds[s + 1]
            gap = np.where(c < lo, lo - c, np.where(c >= hi, c - hi, 0.0))
            need[s] = (gap * gap * (1.0 - 1e-12) <= kth)
        return torch.from_numpy(need)

    def bounded_query(self, xyz, bound2, k):
        q = xyz.shape[1]
        if q == 0:
            return torch.zeros((0, k), dtype=torch.int32), torch.zeros((0, k), dtype=torch.float64)
        idx, d2 = O.knn_bruteforce(self.src, xyz.numpy(), k, gidx=self.gidx)
        out = d2 > bound2.numpy()[:, None]
        idx[out] = NOIDX
        d2[out] = np.inf
        return torch.from_numpy(idx.view(np.int32)), torch.from_numpy(d2)

    def merge(self, idx_lists, d2_lists):
        mi, md = O.merge_candidates(idx_lists.numpy().view(np.uint32), d2_lists.numpy())
        return torch.from_numpy(mi.view(np.int32)), torch.from_numpy(md)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g.load_package()
    from pt_amd import sharding

    n, m, k, seed = 30000, 3000, 8, 0xC4
    case = os.environ.get("PT_CASE", "uniform")
    src = O.synth_xyz(seed, 0, n)
    tgt = O.synth_xyz(seed, 1, m)
    if case == "clustered":           # very uneven slabs: most points near x = 0.8
        src[0] = (0.8 + 0.15 * (src[0] - 0.5)).astype(np.float32)
        bounds = sharding.quantile_slab_bounds(torch.from_numpy(src[0]), world)
    elif case == "tiny":               # fewer points than k in some slabs: lists stay unfilled, bounds are +inf
        src, n = src[:, :10], 10
        bounds = sharding.uniform_slab_bounds(world)
    else:
        bounds = sharding.uniform_slab_bounds(world)
    want_i, want_d = O.knn_bruteforce(src, tgt, k)

    mine_s = np.nonzero((src[0] >= bounds[rank]) & (src[0] < bounds[rank + 1]))[0]
    mine_t = np.nonzero((tgt[0] >= bounds[rank]) & (tgt[0] < bounds[rank + 1]))[0]
    slab = np.ascontiguousarray(src[:, mine_s])
    gidx = mine_s.astype(np.uint32)
    my_tgt = np.ascontiguousarray(tgt[:, mine_t])
    # home answers
    hi, hd = O.knn_bruteforce(slab, my_tgt, k, gidx=gidx)
    idx = torch.from_numpy(hi.view(np.int32).copy())
    d2 = torch.from_numpy(hd.copy())
    comm = sharding.TorchDistComm()
    before = idx.clone()
    changed = []
    sharded_attrs = os.environ.get("PT_ATTRS", "0") == "1"
    if sharded_attrs:
        # the attribute table is sharded with the slabs: this rank holds the records of ITS points only; the home blend is local, the rows
        # completed by other slabs are blended from the records that came with the candidates -- bit-equal to the blend over the whole table
        rgb_all, nrm_all = O.synth_rgb(seed, n), O.synth_nrm(seed, n)
        attrs = sharding.SlabAttributes(torch.from_numpy(gidx.astype(np.int64)), torch.from_numpy(rgb_all[mine_s]), torch.from_numpy(nrm_all[mine_s]))

        def blend_from(idx_t, d2_t, rec):                       # the ORACLE's blend arithmetic on gathered records (a table of c * k rows)
            c_ = idx_t.shape[0]
            fake = np.arange(c_ * k, dtype=np.uint32).reshape(c_, k)
            fake[(idx_t.numpy().view(np.uint32)) == NOIDX] = NOIDX
            flat = rec.reshape(c_ * k, 6).numpy()
            return O.blend(fake, d2_t.numpy(), flat[:, :3].astype(np.uint8), flat[:, 3:].astype(np.float32), mode=0)
        rgb_out, nrm_out = blend_from(idx, d2, attrs.lookup(idx))          # home blend: local records only
        gathered = []

        def changed_cb(rows, rec):
            changed.append(rows.clone()); gathered.append((rows.clone(), rec.clone()))
        st = sharding.exchange_and_merge(comm, OracleSlabEngine(slab, gidx), torch.from_numpy(my_tgt), idx, d2, k, 0, bounds, on_changed=changed_cb, attrs=attrs)
        for rows, rec in gathered:
            r2, n2 = blend_from(idx[rows], d2[rows], rec)
            rgb_out[rows.numpy()] = r2; nrm_out[rows.numpy()] = n2
            pr, pn = sharding.blend_gathered(idx[rows], d2[rows], rec)     # the product's own blend of gathered records: the same within 1e-5
            assert np.abs(pr.numpy() - r2).max(initial=0.0) / 255.0 <= 1e-5 and np.abs(pn.numpy() - n2).max(initial=0.0) <= 1e-5
        want_rgb, want_nrm = O.blend(want_i[mine_t], want_d[mine_t], rgb_all, nrm_all, mode=0)
        attrs_ok = np.array_equal(rgb_out, want_rgb) and np.array_equal(nrm_out, want_nrm)
    else:
        attrs_ok = True
        st = sharding.exchange_and_merge(comm, OracleSlabEngine(slab, gidx), torch.from_numpy(my_tgt), idx, d2, k, 0, bounds,
                                         on_changed=lambda rows: changed.append(rows.clone()))
    got_i = idx.numpy().view(np.uint32)
    ok = attrs_ok and np.array_equal(got_i, want_i[mine_t]) and np.array_equal(d2.numpy(), want_d[mine_t])
    # on_changed names every row whose list was touched (what a fused blend has to redo): rows outside it are unchanged
    touched = torch.zeros(idx.shape[0], dtype=torch.bool)
    for rows in changed:
        touched[rows] = True
    ok = ok and bool((idx[~touched] == before[~touched]).all()) and int(touched.sum()) == st["crossing"]
    # every rank reports; rank 0 aggregates
    flags = [None] * world
    dist.all_gather_object(flags, (bool(ok), len(mine_t), st["crossing"], st["answered"]))
    if rank == 0:
        print("RESULT", flags, flush=True)
        assert all(f[0] for f in flags), flags
        assert sum(f[1] for f in flags) == m
        if case == "uniform":
            assert 0 < sum(f[2] for f in flags) < 0.5 * m      # pruned: only boundary targets cross
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
