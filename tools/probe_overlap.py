"""How much of a memory-bound build hides behind a compute-bound query when both are on the GPU at once?  (dev probe, round 3)
Two contexts with a cloud each (no data dependence between them): context A rebuilds its grid while context B answers its targets,
each from its own host thread on its own stream.  Serial sum vs concurrent wall time = what a slab-pipelined step could gain at most.
    python tools/probe_overlap.py [n] [m]"""
import sys
import threading
import time

sys.path.insert(0, '.')
import torch
import __graft_entry__ as g

pkg = g.load_package()
n, m, k, seed = 500_000_000, 25_000_000, 8, 0xC4
if len(sys.argv) > 2:
    n, m = int(float(sys.argv[1])), int(float(sys.argv[2]))


def now():
    torch.cuda.synchronize()
    return time.perf_counter()


A = pkg.PointsTransfer(device=0, k_hint=k)
B = pkg.PointsTransfer(device=0, k_hint=k)
A.build_synth(n, seed); A.targets_synth(m, seed)
B.build_synth(n, seed + 1); B.targets_synth(m, seed + 1)
dev = "cuda"
idx = torch.empty((m, k), dtype=torch.int32, device=dev); d2 = torch.empty((m, k), dtype=torch.float64, device=dev)
rgb = torch.empty((m, 3), dtype=torch.float32, device=dev); nrm = torch.empty((m, 3), dtype=torch.float32, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
A.set_stream(sA.cuda_stream)


def build_a():
    A.rebuild(); A.synchronize()


def query_b():
    with torch.cuda.stream(sB):
        B.query_blend_resident_dev(k, 0, idx, d2, rgb, nrm)
    sB.synchronize()


for f in (build_a, query_b, build_a, query_b):
    f()
t0 = now(); build_a(); t1 = now(); query_b(); t2 = now()
print("serial: rebuild A %.2f ms, query+blend B %.2f ms, sum %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3), flush=True)
for it in range(4):
    ta, tb = threading.Thread(target=build_a), threading.Thread(target=query_b)
    t0 = now(); ta.start(); tb.start(); ta.join(); tb.join(); t1 = now()
    print("concurrent (two threads, two streams): %.2f ms wall" % ((t1 - t0) * 1e3), flush=True)
# the query started a little later (the build's first phases are the ones a pipelined step could NOT overlap)
for delay in (0.002, 0.005, 0.010):
    ta, tb = threading.Thread(target=build_a), threading.Thread(target=query_b)
    t0 = now(); ta.start(); time.sleep(delay); tb.start(); ta.join(); tb.join(); t1 = now()
    print("concurrent, query started %.0f ms after the build: %.2f ms wall" % (delay * 1e3, (t1 - t0) * 1e3), flush=True)
A.close(); B.close()
