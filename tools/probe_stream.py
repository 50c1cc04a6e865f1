"""pt_stream_query with and without the bounds of round 3 (dev probe): a host-resident cloud streamed in four chunks.
  random : points in random order (every chunk covers the whole volume: nothing to skip, the bound only prunes)
  sorted : the same points in x order (chunks are slabs: most (target, chunk) pairs are out of reach)
Host memory pageable and page-locked (pt_host_alloc).   usage: python tools/probe_stream.py [n] [m]"""
import ctypes as C
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
n, m, k = 400_000_000, 50_000_000, 8
if len(sys.argv) > 2:
    n, m = int(float(sys.argv[1])), int(float(sys.argv[2]))
chunk = n // 4
L = pkg.capi.lib()


def now():
    torch.cuda.synchronize()
    return time.perf_counter()


def pinned(shape, dtype):
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = L.pt_host_alloc(nbytes)
    assert p
    return np.frombuffer((C.c_char * nbytes).from_address(p), dtype=dtype).reshape(shape), p


rng = np.random.default_rng(3)
xyz, hp = pinned((3, n), np.float32)
for a in range(3):
    for s in range(0, n, 50_000_000):
        xyz[a, s:s + 50_000_000] = rng.random(min(50_000_000, n - s), dtype=np.float32)
tgt = rng.random((3, m), dtype=np.float32)
for order in ("random", "sorted"):
    if order == "sorted":
        o = np.argsort(xyz[0], kind="stable")
        for a in range(3):
            xyz[a] = xyz[a][o]
        del o
    for mm in (m // 10, m):
        ref = None
        for bounds in (0, 1):
            if order == "sorted" and bounds == 0 and mm == m:
                continue          # (round 2's behaviour on a cloud in spatial order: 35 s at a tenth of the targets -- once is enough)
            with pkg.PointsTransfer(device=0, k_hint=k) as p:
                p.set_param("stream_bounds", bounds)
                p.set_targets(tgt[:, :mm])
                ts = []
                for rep in range(1 if (order == "sorted" and bounds == 0) else 2):
                    t0 = now(); si, sd = p.stream_query(xyz, chunk, k=k); ts.append(now() - t0)
                st = p.stats()
                sk = "%d, revisited %d" % (st["stream_skipped"], st["stream_revisited"])
            same = None
            if ref is None:
                ref = (si, sd)
            else:
                same = bool(np.array_equal(si, ref[0]) and np.array_equal(sd, ref[1]))
            print("%s order, %d points in 4 chunks, %d targets, k=%d, pinned host memory, bounds %s: %.3f s (%.1f GB/s of coordinates), skipped %s, same as unbounded: %s"
                  % (order, n, mm, k, "on" if bounds else "off", min(ts), n * 12 / min(ts) / 1e9, sk, same), flush=True)
L.pt_host_free(hp)
