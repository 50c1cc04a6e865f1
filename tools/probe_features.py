"""Timings of the round-2 entry points on one GPU (dev probe; the numbers quoted in DESIGN.md sections 7-9):
  bake     : pt_bake_texture on a grid mesh over a uniform cloud (faces, resolution, k from argv)
  stream   : pt_stream_query, host-resident cloud in chunks, against the resident search of the same cloud
  exchange : pt_exchange_merge_local, G logical slabs of a uniform cloud (phase times from the library's stats)
usage: python tools/probe_features.py [bake] [stream] [exchange]"""
import math
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
what = sys.argv[1:] or ["bake", "stream", "exchange"]


def now():
    torch.cuda.synchronize()
    return time.perf_counter()


if "bake" in what:
    n, S, k, R = 20_000_000, 700, 20, 8192
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        rng = np.random.default_rng(7)                                           # the cloud samples the sheet the mesh was decimated from
        src = np.stack([rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32), (0.5 + 1e-4 * rng.standard_normal(n)).astype(np.float32)])
        p.build(src, rgb=rng.integers(0, 256, (n, 3), dtype=np.uint8))
        # a textured sheet through the middle of the cloud: (S+1)^2 vertices, 2 S^2 faces, UVs with different scale per axis
        gx, gy = np.meshgrid(np.arange(S + 1) / S, np.arange(S + 1) / S, indexing="xy")
        verts = np.zeros(((S + 1) ** 2,), dtype=pkg.POINT_DTYPE)
        verts["ver"][:, 0] = 0.02 + 0.96 * gx.ravel(); verts["ver"][:, 1] = 0.02 + 0.96 * gy.ravel(); verts["ver"][:, 2] = 0.5
        verts["U"] = 0.013 + 0.971 * gx.ravel(); verts["V"] = 0.021 + 0.953 * gy.ravel()
        vid = (np.arange(S)[:, None] * (S + 1) + np.arange(S)[None, :]).ravel()
        faces = np.concatenate([np.stack([vid, vid + 1, vid + S + 2], 1), np.stack([vid, vid + S + 2, vid + S + 1], 1)]).astype(np.int32)
        t0 = now(); idx, _ = p.query(np.ascontiguousarray(verts["ver"].T, dtype=np.float32), k=k); t1 = now()
        for rep in range(2):
            t2 = now(); tex = p.bake_texture(verts, faces, idx, resolution=R, pad_ksize=25); t3 = now()
            st = p.stats()
            print("bake: %d faces, k=%d, %d^2 atlas: query (host path) %.1f ms, pt_bake_texture wall %.1f ms (device %.2f ms), covered %.1f %%" %
                  (faces.shape[0], k, R, (t1 - t0) * 1e3, (t3 - t2) * 1e3, st["ms_bake"], 100.0 * float((tex[:, :, 3] > 0).mean())), flush=True)

if "stream" in what:
    n, m, k, chunk = 400_000_000, 5_000_000, 8, 100_000_000
    rng = np.random.default_rng(3)
    xyz = np.empty((3, n), np.float32)
    for a in range(3):
        for s in range(0, n, 50_000_000):
            xyz[a, s:s + 50_000_000] = rng.random(min(50_000_000, n - s), dtype=np.float32)
    tgt = rng.random((3, m), dtype=np.float32)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_targets(tgt)
        for rep in range(2):
            t0 = now(); si, sd = p.stream_query(xyz, chunk, k=k); t1 = now()
            print("stream: %d points in chunks of %d, %d targets, k=%d: %.2f s (%.1f GB/s of coordinates, %.1f M targets/s)" %
                  (n, chunk, m, k, t1 - t0, n * 12 / (t1 - t0) / 1e9, m / (t1 - t0) / 1e6), flush=True)
        t0 = now(); p.build(xyz); t1 = now(); ri, rd = p.query(tgt, k=k); t2 = now()
        print("resident: build from host %.2f s, query (host path) %.2f s; streamed == resident: %s" %
              (t1 - t0, t2 - t1, bool(np.array_equal(si, ri.astype(np.uint64)) and np.array_equal(sd, rd))), flush=True)
    del xyz

if "exchange" in what:
    G, n, m, k = 4, 400_000_000, 20_000_000, 8
    bounds = [-math.inf] + [s / G for s in range(1, G)] + [math.inf]
    pts, xs, ii, dd, cc, nn = [], [], [], [], [], []
    for s in range(G):
        p = pkg.PointsTransfer(device=0, k_hint=k)
        lo, hi = bounds[s], bounds[s + 1]
        p.build_synth(n, 0xC4, slab_axis=0, slab_lo=lo, slab_hi=hi)
        p.targets_synth(m, 0xC4, slab_axis=0, slab_lo=lo, slab_hi=hi)
        ms = p.num_targets
        x = torch.empty((3, ms), dtype=torch.float32, device="cuda")
        i_ = torch.empty((ms, k), dtype=torch.int32, device="cuda"); d_ = torch.empty((ms, k), dtype=torch.float64, device="cuda")
        c_ = torch.empty((ms, 3), dtype=torch.float32, device="cuda"); n_ = torch.empty((ms, 3), dtype=torch.float32, device="cuda")
        p.query_blend_resident_dev(k, 0, i_, d_, c_, n_)
        p.resident_target_xyz_dev(x)
        pts.append(p); xs.append(x); ii.append(i_); dd.append(d_); cc.append(c_); nn.append(n_)
    for rep in range(3):
        if rep:
            for s in range(G):
                pts[s].query_blend_resident_dev(k, 0, ii[s], dd[s], cc[s], nn[s])
        t0 = now()
        pkg.PointsTransfer.exchange_merge_local(pts, xs, pkg.F32, k, 0, bounds, ii, dd, pkg.BLEND_MEAN, cc, nn)
        t1 = now()
        print("exchange: G=%d logical slabs of a %d-point cloud, %d targets: %.2f ms wall for all %d slabs (%.2f ms per slab)" %
              (G, n, m, (t1 - t0) * 1e3, G, (t1 - t0) * 1e3 / G), flush=True)
    for p in pts:
        p.close()
