"""Does the blend's gather traffic hide behind the tile kernel when both run at once (two streams)?  Dev probe at C4.
Measures: fused k-NN + blend; k-NN alone; blend alone; k-NN on one stream with the blend of a previous result on another."""
import sys
import time

sys.path.insert(0, '.')
import torch
import __graft_entry__ as g

pkg = g.load_package()
n, m, k, seed = 1_000_000_000, 50_000_000, 8, 0xC4
if len(sys.argv) > 1:
    n, m = int(float(sys.argv[1])), int(float(sys.argv[2]))


def now():
    torch.cuda.synchronize()
    return time.perf_counter()


with pkg.PointsTransfer(device=0, k_hint=k) as p:
    p.build_synth(n, seed); p.targets_synth(m, seed)
    dev = "cuda"
    idx = torch.empty((m, k), dtype=torch.int32, device=dev); d2 = torch.empty((m, k), dtype=torch.float64, device=dev)
    rgb = torch.empty((m, 3), dtype=torch.float32, device=dev); nrm = torch.empty((m, 3), dtype=torch.float32, device=dev)
    p.rebuild()
    for it in range(2):
        t0 = now(); p.query_blend_resident_dev(k, 0, idx, d2, rgb, nrm); t1 = now()
    print("fused k-NN + blend: %.2f ms wall" % ((t1 - t0) * 1e3), flush=True)
    for it in range(2):
        t0 = now(); p.query_resident_dev(k, idx, d2); t1 = now()
    print("k-NN alone: %.2f ms wall" % ((t1 - t0) * 1e3), flush=True)
    idx2, d22 = idx.clone(), d2.clone()
    rgb2 = torch.empty_like(rgb); nrm2 = torch.empty_like(nrm)
    for it in range(2):
        t0 = now(); p.blend_dev(idx2, d22, m, k, 0, rgb2, nrm2); t1 = now()
    print("blend alone: %.2f ms wall" % ((t1 - t0) * 1e3), flush=True)
    s1, s2 = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)      # the tile kernel's queue is served first
    print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
    for it in range(3):
        t0 = now()
        with torch.cuda.stream(s2):
            p.blend_dev(idx2, d22, m, k, 0, rgb2, nrm2)          # asynchronous
        with torch.cuda.stream(s1):
            p.query_resident_dev(k, idx, d2)                      # returns when the tile kernel's leftovers are known
        t1 = now()
        print("k-NN (stream 1) + blend of another result (stream 2): %.2f ms wall" % ((t1 - t0) * 1e3), flush=True)
    # halves: blend of half A while half B is searched would look like this in time
    h = m // 2
    for it in range(2):
        t0 = now()
        with torch.cuda.stream(s2):
            p.blend_dev(idx2[:h], d22[:h], h, k, 0, rgb2[:h], nrm2[:h])
        with torch.cuda.stream(s1):
            p.query_resident_dev(k, idx, d2)
        t1 = now()
        print("k-NN (all) + blend of HALF the rows on stream 2: %.2f ms wall" % ((t1 - t0) * 1e3), flush=True)

    # the other order: the tile kernel is already on the GPU when the blend arrives (k-NN launched from a helper thread)
    import threading
    for delay in (0.0005, 0.002, 0.005):
        for it in range(2):
            def knn():
                with torch.cuda.stream(s1):
                    p.query_resident_dev(k, idx, d2)
            t0 = now()
            th = threading.Thread(target=knn); th.start()
            time.sleep(delay)
            with torch.cuda.stream(s2):
                p.blend_dev(idx2, d22, m, k, 0, rgb2, nrm2)
            th.join()
            t1 = now()
            print("k-NN first, blend of another result %.1f ms later on stream 2: %.2f ms wall" % (delay * 1e3, (t1 - t0) * 1e3), flush=True)
