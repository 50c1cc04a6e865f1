"""Spatial-slab sharding of the detail-transfer path across GPUs (one process per GPU).

The reference is a single process; all of this is new design, following SURVEY.md 8(e):

  * the source cloud is cut into G slabs along one axis at equal-count quantiles; rank g holds slab g, builds its
    own grid over it, and keeps ORIGINAL (global) point indices;
  * every target has a HOME slab (the one its coordinate falls in); the home rank answers it first;
  * a target needs another slab s only if dist2(target, slab s) <= its current k-th squared distance -- the
    reference's Distance::min_distance_to_rectangle (src/Distance.h:27-57) applied to slab boxes;
  * the (few) targets that need other slabs are exchanged with ONE all-gather of request packets, answered by the
    owning ranks with a radius-bounded search, and returned with ONE all-gather of candidate packets; the home
    rank merges its own k with the returned k's under the total order (d2, index).
    north_star: "each GPU returning its local k candidates with an RCCL allgather over xGMI to merge";
  * (round 4) the attribute table may be SHARDED with the slabs (SlabAttributes): the answers then carry their
    candidates' records and the owner blends the completed rows from what it gathered -- per-GPU memory falls with G.

Collectives are torch.distributed all_gather (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).  The
compute steps go through a small engine interface so that the CPU tests can drive the very same protocol code
with an oracle-backed engine; the GPU engine below calls libpt_hip.so only.
"""
import math

import torch

NOIDX = 0xFFFFFFFF


# ---- collectives -------------------------------------------------------------------------------------------
class TorchDistComm:
    """torch.distributed front-end: the only thing the protocol needs is a same-shape all_gather and a max."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._d = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_gather(self, t):
        # concatenated form (world * rows, ...): accepted by both RCCL and gloo; viewed as [world, rows, ...]
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self._d.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out.view((self.world,) + tuple(t.shape))

    def max_int(self, v, device):
        t = torch.tensor([int(v)], dtype=torch.int64, device=device)
        self._d.all_reduce(t, op=self._d.ReduceOp.MAX, group=self.group)
        return int(t.item())


class HostStagedComm(TorchDistComm):
    """Same interface over a backend that cannot move device tensors (gloo): collectives are staged through host
    memory.  Used to rehearse the multi-rank bench flow with several ranks sharing ONE GPU (RCCL refuses that)."""

    def all_gather(self, t):
        h = t.detach().cpu().contiguous()
        out = torch.empty((self.world * h.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype)
        self._d.all_gather_into_tensor(out, h, group=self.group)
        return out.view((self.world,) + tuple(h.shape)).to(t.device)

    def max_int(self, v, device):
        return super().max_int(v, "cpu")


class SingleComm:
    """world_size 1 (no exchange ever happens)."""
    rank, world = 0, 1

    def all_gather(self, t):
        return t.unsqueeze(0)

    def max_int(self, v, device):
        return int(v)


# ---- slab geometry -----------------------------------------------------------------------------------------
def uniform_slab_bounds(world, lo=0.0, hi=1.0):
    """Equal-count quantiles of a uniform cloud on [lo, hi): G+1 ascending bounds, open-ended at both ends."""
    b = [lo + (hi - lo) * g / world for g in range(world + 1)]
    b[0], b[-1] = -math.inf, math.inf
    return b


def quantile_slab_bounds(coords, world):
    """Equal-count quantiles from a (sample of the) cloud's coordinates along the slab axis (1-D tensor)."""
    q = torch.quantile(coords.double().cpu(), torch.arange(1, world, dtype=torch.float64) / world) if world > 1 else torch.empty(0)
    return [-math.inf] + [float(v) for v in q] + [math.inf]


# ---- engines -------------------------------------------------------------------------------------------------
class GpuSlabEngine:
    """The compute steps of the protocol on the GPU, through the C ABI (PointsTransfer)."""

    def __init__(self, pt, xyz_type, device):
        from . import capi
        self.pt, self.xyz_type, self.device = pt, xyz_type, device
        self.tdtype = torch.float64 if xyz_type == capi.F64 else torch.float32

    def slab_need(self, xyz, d2, k, axis, bounds, my_slab):
        m = xyz.shape[1]
        need = torch.empty((len(bounds) - 1, m), dtype=torch.uint8, device=self.device)
        if m:
            self.pt.slab_need_dev(xyz, self.xyz_type, d2, m, k, axis, bounds, my_slab, need)
        return need

    def pack_requests(self, xyz, d2, k, axis, bounds, my_slab):
        """slab_need and the selection of the crossing targets fused on the device: (rows [c] int64, packets [c, 5] f64)."""
        m = xyz.shape[1]
        if getattr(self, "_req_sel", None) is None or self._req_cap < m:   # scratch sized for every target once, reused by later
            # steps -- allocated on the first call even when this rank holds NO targets (a mesh that covers only part of the
            # slab axis): the empty views returned below must exist, or this rank dies before the collective the others wait in
            self._req_sel = torch.empty((m,), dtype=torch.int32, device=self.device)
            self._req_pkt = torch.empty((m, 5), dtype=torch.float64, device=self.device)
            self._req_cap = m
        c = self.pt.pack_requests_dev(xyz, self.xyz_type, d2, m, k, axis, bounds, my_slab, self._req_sel, self._req_pkt) if m else 0
        return self._req_sel[:c].to(torch.int64), self._req_pkt[:c]

    def bounded_query(self, xyz, bound2, k):
        c = xyz.shape[1]
        idx = torch.empty((c, k), dtype=torch.int32, device=self.device)
        d2 = torch.empty((c, k), dtype=torch.float64, device=self.device)
        if c:
            self.pt.query_bounded_dev(xyz.contiguous(), self.xyz_type, bound2.contiguous(), c, k, idx, d2)
        return idx, d2

    def merge(self, idx_lists, d2_lists):
        g, c, k = idx_lists.shape
        idx = torch.empty((c, k), dtype=torch.int32, device=self.device)
        d2 = torch.empty((c, k), dtype=torch.float64, device=self.device)
        if c:
            self.pt.merge_candidates_dev(idx_lists.contiguous(), d2_lists.contiguous(), g, c, k, idx, d2)
        return idx, d2


# ---- attributes sharded with the slabs (round 4) --------------------------------------------------------------------------
class SlabAttributes:
    """The attribute records of ONE slab's points, kept where the points are (1/G of the table per GPU instead of a copy of all of
    it: 16 GB at 1e9 points).  A home search only ever names points of the home slab, so its blend is local; what another slab
    contributes arrives WITH its candidates (exchange_and_merge(..., attrs=...)): 24 bytes per candidate for the ~2 % of the targets
    that cross a slab border.  gidx: the slab's global point indices (any order); rgb [n, 3] u8, nrm [n, 3] f32 in the same order."""

    def __init__(self, gidx, rgb, nrm):
        g = torch.as_tensor(gidx).to(torch.int64) & 0xFFFFFFFF
        self.sorted_gidx, self.perm = torch.sort(g)
        self.table = torch.cat([torch.as_tensor(rgb).to(torch.float64), torch.as_tensor(nrm).to(torch.float64)], dim=1).to(g.device)   # u8 / f32 are exact in f64

    def lookup(self, idx_i32):
        """[q, k] indices (u32 bit patterns; NOIDX = empty) -> [q, k, 6] records (zeros for empty entries).  Every named point must
        belong to this slab."""
        u = idx_i32.to(torch.int64) & 0xFFFFFFFF
        valid = u != NOIDX
        pos = torch.searchsorted(self.sorted_gidx, torch.where(valid, u, self.sorted_gidx[:1].expand_as(u)) if self.sorted_gidx.numel() else u)
        pos = pos.clamp(max=max(int(self.sorted_gidx.numel()) - 1, 0))
        if self.sorted_gidx.numel():
            assert bool((self.sorted_gidx[pos][valid] == u[valid]).all()), "a home list names a point of another slab"
            rec = self.table[self.perm[pos]]
        else:
            rec = torch.zeros(tuple(u.shape) + (6,), dtype=torch.float64, device=u.device)
        return torch.where(valid.unsqueeze(-1), rec, torch.zeros_like(rec))


def blend_gathered(idx_i32, d2, rec, mode=0):
    """The blend of pt_attr.hip's blend_one from GATHERED records: rec [c, k, 6] = (r, g, b, nx, ny, nz) of the k neighbours in list
    order, fp64 sums in that order, then one normalisation.  Returns (rgb [c, 3] f32, nrm [c, 3] f32)."""
    valid = ((idx_i32.to(torch.int64) & 0xFFFFFFFF) != NOIDX)
    w = torch.where(valid, (1.0 / (d2 + 1e-12)) if mode == 1 else torch.ones_like(d2), torch.zeros_like(d2))
    acc = torch.zeros((rec.shape[0], 6), dtype=torch.float64, device=rec.device)
    wsum = torch.zeros((rec.shape[0],), dtype=torch.float64, device=rec.device)
    for j in range(rec.shape[1]):                                              # left to right, like the kernel
        acc = acc + w[:, j:j + 1] * rec[:, j]
        wsum = wsum + w[:, j]
    has = wsum > 0
    acc = torch.where(has.unsqueeze(1), acc / torch.where(has, wsum, torch.ones_like(wsum)).unsqueeze(1), acc)
    nn = acc[:, 3:]
    ln = torch.sqrt((nn * nn).sum(dim=1))
    nn = torch.where((ln >= 1e-12).unsqueeze(1), nn / torch.where(ln >= 1e-12, ln, torch.ones_like(ln)).unsqueeze(1), nn)
    return acc[:, :3].to(torch.float32), nn.to(torch.float32)


# ---- the protocol ------------------------------------------------------------------------------------------------
def _idx_to_f64(idx_i32):
    return (idx_i32.to(torch.int64) & 0xFFFFFFFF).to(torch.float64)        # u32 values are exact in f64


def _f64_to_idx(v):
    u = v.to(torch.int64)
    return torch.where(u >= 2**31, u - 2**32, u).to(torch.int32)          # back to the u32 bit pattern


def exchange_and_merge(comm, engine, xyz, idx, d2, k, axis, bounds, on_changed=None, attrs=None):
    """Complete the home-slab answers (idx int32 [m,k] holding u32 bit patterns, d2 f64 [m,k], updated IN PLACE)
    with the candidates of the other slabs.  xyz: [3,m] planar coordinates of this rank's targets.
    on_changed(rows): called with the local rows whose lists were merged with foreign candidates (anything derived from
    the home-slab lists of those rows -- a fused blend -- has to be redone for them).
    attrs (a SlabAttributes over THIS rank's slab, or None): the attribute table is sharded with the slabs -- every answer then
    carries its candidates' records, the merge carries them along, and on_changed(rows, rec) receives rec [c, k, 6], the records of
    the merged lists' neighbours in list order (blend_gathered turns them into the blend of those rows).
    Returns counters: targets sent out, foreign targets answered, bytes all-gathered per rank."""
    G, me = comm.world, comm.rank
    stats = {"crossing": 0, "answered": 0, "bytes_gathered": 0}
    if G == 1:
        return stats
    assert G <= 52, "the need bitmask travels as an exact integer in one f64"
    dev = idx.device
    m = xyz.shape[1]
    # -- request packets: x, y, z, bound (current k-th d2), slab bitmask ------------------------------------------
    if hasattr(engine, "pack_requests"):                                     # one device pass (GPU engine)
        sel, mine_pkt = engine.pack_requests(xyz, d2, k, axis, bounds, me)
        c = int(sel.numel())
    else:                                                                    # the same in torch ops (oracle-backed engine of the CPU tests)
        need = engine.slab_need(xyz, d2, k, axis, bounds, me)                 # [G, m] u8
        sel = torch.nonzero(need.any(dim=0)).flatten()                       # my targets that need another slab
        c = int(sel.numel())
        weights = (2.0 ** torch.arange(G, dtype=torch.float64, device=dev)).unsqueeze(1)
        mine_pkt = torch.empty((c, 5), dtype=torch.float64, device=dev)
        if c:
            mine_pkt[:, 0:3] = xyz[:, sel].t().to(torch.float64)
            mine_pkt[:, 3] = d2[sel, k - 1]
            mine_pkt[:, 4] = (need[:, sel].to(torch.float64) * weights).sum(dim=0)
    cmax = comm.max_int(c, dev)
    stats["crossing"] = c
    if cmax == 0:
        return stats
    pkt = torch.zeros((cmax, 5), dtype=torch.float64, device=dev)
    if c:
        pkt[:c] = mine_pkt
    allreq = comm.all_gather(pkt)                                             # [G, cmax, 5]
    stats["bytes_gathered"] += allreq.numel() * 8
    # -- answer the requests addressed to my slab ---------------------------------------------------------------
    mask = allreq[:, :, 4].to(torch.int64)
    mine = ((mask >> me) & 1).bool()
    mine[me] = False
    who = torch.nonzero(mine)                                                 # [q, 2] = (owner rank, row in its packet)
    q = int(who.shape[0])
    stats["answered"] = q
    rows = allreq[who[:, 0], who[:, 1]] if q else torch.zeros((0, 5), dtype=torch.float64, device=dev)
    qxyz = rows[:, 0:3].t().contiguous().to(xyz.dtype)                        # exact: they were widened from this dtype
    ai, ad = engine.bounded_query(qxyz, rows[:, 3].contiguous(), k)
    qmax = comm.max_int(q, dev)
    if qmax == 0:            # cannot happen when cmax > 0, but never all-gather an empty tensor
        return stats
    aw = 6 * k if attrs is not None else 0                                  # (sharded attributes: the candidates' records ride along)
    ans = torch.full((qmax, 2 + 2 * k + aw), -1.0, dtype=torch.float64, device=dev)
    if q:
        ans[:q, 0] = who[:, 0].to(torch.float64)
        ans[:q, 1] = who[:, 1].to(torch.float64)
        ans[:q, 2:2 + k] = ad
        ans[:q, 2 + k:2 + 2 * k] = _idx_to_f64(ai)
        if aw:
            ans[:q, 2 + 2 * k:] = attrs.lookup(ai).reshape(q, aw)
    allans = comm.all_gather(ans)                                             # [G, qmax, 2+2k]
    stats["bytes_gathered"] += allans.numel() * 8
    # -- merge what came back for my targets ----------------------------------------------------------------------
    if c == 0:
        return stats
    li = torch.full((G, c, k), -1, dtype=torch.int32, device=dev)             # -1 = 0xFFFFFFFF = empty
    ld = torch.full((G, c, k), math.inf, dtype=torch.float64, device=dev)
    li[me] = idx[sel]
    ld[me] = d2[sel]
    back = torch.nonzero(allans[:, :, 0] == float(me))                       # (server rank, row)
    la = None
    if attrs is not None:
        la = torch.zeros((G, c, k, 6), dtype=torch.float64, device=dev)
        la[me] = attrs.lookup(idx[sel])                                       # the home lists name home points only
    if back.numel():
        rec = allans[back[:, 0], back[:, 1]]
        prow = rec[:, 1].to(torch.int64)
        li[back[:, 0], prow] = _f64_to_idx(rec[:, 2 + k:2 + 2 * k])
        ld[back[:, 0], prow] = rec[:, 2:2 + k]
        if la is not None:
            la[back[:, 0], prow] = rec[:, 2 + 2 * k:].reshape(-1, k, 6)
    mi, md = engine.merge(li, ld)
    idx[sel] = mi
    d2[sel] = md
    if la is None:
        if on_changed is not None:
            on_changed(sel)
        return stats
    # the records of the merged lists, in list order: the G * k candidates of every row ordered by (d2, index) -- two stable sorts --
    # and cut at k; the order must be the merge's own (checked: the blend's sums run in list order)
    ci = (li.permute(1, 0, 2).reshape(c, G * k).to(torch.int64)) & 0xFFFFFFFF
    cd = ld.permute(1, 0, 2).reshape(c, G * k)
    ca = la.permute(1, 0, 2, 3).reshape(c, G * k, 6)
    o1 = torch.sort(ci, dim=1, stable=True).indices
    o2 = torch.sort(torch.gather(cd, 1, o1), dim=1, stable=True).indices
    order = torch.gather(o1, 1, o2)[:, :k]
    assert bool((torch.gather(ci, 1, order) == (mi.to(torch.int64) & 0xFFFFFFFF)).all()), "attribute merge out of step with the list merge"
    ma = torch.gather(ca, 1, order.unsqueeze(-1).expand(c, k, 6))
    if on_changed is not None:
        on_changed(sel, ma)
    return stats
