#!/usr/bin/env python3
"""bench.py -- headline benchmark of the detail-transfer hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
    grid build over the source cloud  ->  target binning  ->  k-NN search  ->  [slab exchange + merge]  ->  blend
Metric (BASELINE.json): target points/sec (k=8 detail transfer); value = targets of the WHOLE job / time.
Default workload = BASELINE config 4, the one north_star quotes the target on: 1B-point source / 50M targets /
k=8, uniform fp32, seed 0xC4 (fits one MI355X: ~63 GB).  With --gpus N the same cloud is sharded by spatial
slab over N ranks (strong scaling: total work fixed), one process per GPU, RCCL via torch.distributed.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|C3|C4|C5] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) the process becomes a LAUNCHER: before torch is imported or
any GPU call made it starts N fresh rank processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT),
relays rank 0's JSON line, stops every rank as soon as one fails, and exits with the first failure's code.  Under
torch.distributed.run the ranks are the launcher's and this file only reads the environment.

Prints ONE JSON line on rank 0 (contract fields + "roofline" + "cpu_baseline" + per-phase detail).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (N source, M target, k, seed)  -- SURVEY.md 8(d); C5 is clustered + fp16 (see WORKLOAD_KIND)
    "C1": (10_000, 1_000, 1, 0xC1),
    "C2": (10_000_000, 1_000_000, 8, 0xC2),
    "C3": (100_000_000, 10_000_000, 16, 0xC3),
    "C4": (1_000_000_000, 50_000_000, 8, 0xC4),
    "C5": (1_000_000_000, 50_000_000, 32, 0xC5),
}
WORKLOAD_KIND = {"C5": ("clustered", "f16")}      # everything else: uniform, fp32
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
KERNEL_NAMES = ["bbox_reduce", "pass1_histogram", "pass1_scatter", "pass2_histogram_scan", "pass2_scatter", "finalize_cellsort",
                "target_sort", "knn_query"]


def survey_alg_bytes(name, n, m, k, s=12, pooled1=True, pooled2=True):
    """SURVEY.md 8(d)'s algorithmic bytes of the phase a kernel belongs to -- the figure `roofline.achieved` / `frac` are
    quoted on.  Query (k-NN + blend): N*s + M*s + M*k*16 + M*(4k + 24); build: N*(2s + 4), which 8(d) gives for the whole
    build and which is split here over its data-moving passes in proportion to the bytes each must move -- the passes that RAN
    (pooled1 / pooled2: `pt_stats.pass1_pooled` / `pass2_pooled` of the timed steps): the exact pass 1 reads the cloud once more
    for its histogram (12 B/point) where the pooled one reads a 1/64 sample, the exact pass 2 reads 2-byte block ids."""
    if name == "knn_query":
        return n * s + m * s + m * k * 16 + m * (4 * k + 24)
    share = {"bbox_reduce": 0.0, "pass1_histogram": 0.2 if pooled1 else 12, "pass1_scatter": 28, "pass2_histogram_scan": 0.0 if pooled2 else 2, "pass2_scatter": 32, "finalize_cellsort": 32}
    if name in share:
        return n * (2 * s + 4) * share[name] / sum(share.values())
    return m * (2 * s + 4)            # target_sort: the same formula over the targets


def kernel_alg_bytes(name, n, m, k, s=12, pooled1=True, pooled2=True):
    """The IMPLEMENTATION's own minimum for one launch of that kernel: inputs read once + outputs
    written once with the record sizes this build uses.  s = bytes per xyz (12 for fp32); records are s+4 (xyz + original
    index).  Reported as `impl_bytes_per_launch`, never as the roofline fraction."""
    rec = s + 4
    return {
        "bbox_reduce": n * s,
        "pass1_histogram": n * s // 64 if pooled1 else n * s,        # (the pooled pass 1 reads a 1/64 sample; the exact one the cloud)
        "pass1_scatter": n * (s + rec),
        "pass2_histogram_scan": 0 if pooled2 else n * 2,             # (2-byte block ids; nothing when pass 2 is pooled)
        "pass2_scatter": n * 2 * rec,
        "finalize_cellsort": n * 2 * rec,
        "target_sort": m * (s + rec),
        # scan every source record once, read targets, write idx + d2; fused blend: gather k 16-B attribute records, write rgb + normal
        "knn_query": n * rec + m * rec + m * k * 12 + m * k * 16 + m * 24,
    }[name]


# (since round 3 big clouds run the pooled pass 1: pool_sample_kernel instead of the histogram pass, scatter_pool_kernel instead of
#  scatter_chunk_kernel; the traffic summary is looked up under either name)
PMC_KERNEL = {"bbox_reduce": "bbox_kernel", "pass1_histogram": ("pool_sample_kernel", "hist_chunk_kernel"), "pass1_scatter": ("scatter_pool_kernel", "scatter_chunk_kernel"),
              "pass2_histogram_scan": "hist_bid_kernel", "pass2_scatter": "scatter_kernel", "finalize_cellsort": "finalize_kernel",
              "knn_query": "knn_tile_kernel"}


def pmc_profile(workload, world):
    """The newest committed PMC traffic summary of this workload (profiles/rNN_*_<workload>_pmc_traffic.json, written by
    tools/pmc_traffic.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950).  Returns (dict, file name) or (None, None)."""
    import glob
    if world != 1:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*%s_pmc_traffic.json" % workload.lower())))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.basename(files[-1])


def pmc_traffic(kernel, workload, world):
    """HBM bytes of one launch of `kernel` (or of a whole step, "__step__") from that summary; None when there is none."""
    prof, _ = pmc_profile(workload, world)
    if prof is None:
        return None
    if kernel == "__step__":
        return prof.get("step", {}).get("hbm_bytes")
    names = PMC_KERNEL.get(kernel, "")
    for name in (names if isinstance(names, tuple) else (names,)):
        rec = prof["kernels"].get(name)
        if rec:
            return rec["hbm_bytes"]
    return None


def cpu_baseline(n_total, m_total, k, seed):
    """The oracle's kd-tree restatement of the reference's CPU search (kind "port"), timed on this host's cores on a
    bounded sample with the workload's N/M ratio.  Reported beside the GPU number -- never the target."""
    from oracle import oracle as O
    O.build()
    ratio = max(1, n_total // max(m_total, 1))
    ns = min(n_total, 40_000_000)     # ~10-30 s of CPU work on the GPU box's host (the build is single-threaded)
    ms = max(1000, min(m_total, ns // ratio))
    src = O.synth_xyz(seed, 0, ns)
    tgt = O.synth_xyz(seed, 1, ms)
    cores = O.num_threads()                  # before the 1-thread run below pins OpenMP to one thread
    t0 = time.perf_counter()
    kd = O.KdTree(src)                       # single-threaded, like the reference's CGAL build
    t1 = time.perf_counter()
    kd.query(tgt, k)                         # OpenMP over targets, all host cores
    t2 = time.perf_counter()
    m1 = max(1000, ms // 40)
    kd.query(tgt[:, :m1], k, nthreads=1)     # the reference's default build is single-threaded (src/CMakeLists.txt:26)
    t3 = time.perf_counter()
    kd.close()
    return {
        "value": ms / (t2 - t0), "unit": "target points/sec", "cores": cores, "kind": "port",
        "sample": "first %d of %d source points and first %d of %d targets (same N/M ratio, same generator/seed), k=%d; "
                  "kd-tree build %.2f s on 1 thread + query %.3f s on %d threads; value = sample targets / (build + query)"
                  % (ns, n_total, ms, m_total, k, t1 - t0, t2 - t1, cores),
        "query_only_value": ms / (t2 - t1),
        "query_only_1thread_value": m1 / (t3 - t2),
    }


class BoxSampler:
    """What the GPU this rank runs on is SET to and what it DOES while the timed steps run (the same binary is 49.5 ms on
    one MI355X and 54.2 on another -- the JSON line now says what the box looked like).  The card is found in sysfs by PCI address
    (torch's device properties; no extra GPU call); a host thread reads shader clock, socket power, temperatures and the busy
    percentages every few milliseconds between start() and stop().  Host-side file reads only: nothing is added to the GPU's queue."""

    def __init__(self, torch, index):
        import glob
        self.dir = self.hwmon = None
        self.props = {}
        try:
            pr = torch.cuda.get_device_properties(index)
            self.props = {"name": pr.name, "arch": getattr(pr, "gcnArchName", None), "compute_units": pr.multi_processor_count,
                          "l2_bytes": getattr(pr, "L2_cache_size", None), "hbm_bytes": pr.total_memory}
            want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
                if "-" in os.path.basename(card):
                    continue
                if os.path.basename(os.path.realpath(os.path.join(card, "device"))).lower().startswith(want):
                    self.dir = os.path.join(card, "device")
                    hw = glob.glob(os.path.join(self.dir, "hwmon", "hwmon*"))
                    self.hwmon = hw[0] if hw else None
        except Exception:           # noqa: BLE001 -- a box that hides sysfs just reports nothing
            pass
        self.samples = []
        self._stop = None
        self._thread = None

    @staticmethod
    def _rd(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None

    def _num(self, base, name, scale):
        v = self._rd(os.path.join(base, name)) if base else None
        try:
            return float(v) * scale
        except (TypeError, ValueError):
            return None

    def static(self):
        if not self.dir:
            return dict(self.props, card=None)
        level = lambda f: next((l.split(":")[1].strip().rstrip("*").strip() for l in (self._rd(os.path.join(self.dir, f)) or "").splitlines() if l.endswith("*")), None)
        return dict(self.props, **{"card": os.path.basename(os.path.dirname(self.dir)), "pci": os.path.basename(os.path.realpath(self.dir)),
                "compute_partition": self._rd(os.path.join(self.dir, "current_compute_partition")),
                "memory_partition": self._rd(os.path.join(self.dir, "current_memory_partition")),
                "perf_level": self._rd(os.path.join(self.dir, "power_dpm_force_performance_level")),
                "sclk_levels": (self._rd(os.path.join(self.dir, "pp_dpm_sclk")) or "").replace("\n", " | "),
                "mclk_mhz": level("pp_dpm_mclk"), "fclk_mhz": level("pp_dpm_fclk"),
                "power_cap_w": self._num(self.hwmon, "power1_cap", 1e-6), "vbios": self._rd(os.path.join(self.dir, "vbios_version"))})

    def _one(self):
        return (self._num(self.hwmon, "freq1_input", 1e-6), self._num(self.hwmon, "power1_input", 1e-6), self._num(self.hwmon, "temp2_input", 1e-3),
                self._num(self.hwmon, "temp3_input", 1e-3), self._num(self.dir, "gpu_busy_percent", 1.0), self._num(self.dir, "mem_busy_percent", 1.0))

    def start(self, period=0.004):
        if not self.dir:
            return
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                self.samples.append(self._one())
                self._stop.wait(period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread:
            self._stop.set()
            self._thread.join()

    def summary(self):
        out = self.static()
        cols = ("sclk_mhz", "power_w", "temp_junction_c", "temp_mem_c", "gpu_busy_pct", "mem_busy_pct")
        for i, name in enumerate(cols):
            v = [t[i] for t in self.samples if t[i] is not None]
            out[name] = {"min": min(v), "mean": round(sum(v) / len(v), 1), "max": max(v)} if v else None
        out["samples"] = len(self.samples)
        return out


def dry_run(args, world, rank, dist, torch):
    """--dry-run: everything of a multi-rank run EXCEPT the GPU work -- rendezvous, the fence, K "steps" (a sleep), the max over ranks,
    one JSON line from rank 0 -- so that the launcher and the collectives' plumbing are testable on a machine without GPUs."""
    if world > 1:
        dist.init_process_group(args.backend if args.backend == "gloo" else "gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"dry_run": True, "metric": "none (launcher / rendezvous rehearsal, no GPU work)", "value": None, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / max(args.steps, 1) * 1e3}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(n, argv):
    """The launcher of a plain `bench.py --gpus N`: N fresh children (never an exec, never a fork of a process that touched the
    GPU -- this one has not even imported torch), rank 0's stdout relayed, everybody else's sent to stderr so that stdout carries
    ONE JSON line; the first rank that fails takes the others down (they would wait for it in a collective) and its exit code is
    the launcher's."""
    import signal
    import socket
    import subprocess
    with socket.socket() as so:                      # a free rendezvous port on the loop-back interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))

    def relay():                                     # rank 0's JSON line(s) to stdout, anything else it prints (library banners) to stderr
        for raw in procs[0].stdout:
            line = raw.decode(errors="replace")
            out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            out.write(line)
            out.flush()
    import threading
    pump = threading.Thread(target=relay, daemon=True)
    pump.start()
    worst, live = 0, set(range(n))
    try:
        while live and not worst:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is not None:
                    live.discard(r)
                    if rc and not worst:
                        worst = rc if rc > 0 else 128 - rc
                        print("bench.py: rank %d exited with code %d: stopping the other %d rank(s)" % (r, rc, len(live)), file=sys.stderr)
            time.sleep(0.05)
    except KeyboardInterrupt:
        worst = 130
    if live:
        for r in live:
            procs[r].send_signal(signal.SIGTERM)
        t0 = time.time()
        for r in sorted(live):
            try:
                procs[r].wait(timeout=max(0.1, 10.0 - (time.time() - t0)))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    pump.join(timeout=5)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # (a C4 step is ~50 ms: five steps ended before the clocks had settled)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal only: ranks may share one GPU, collectives staged through host memory")
    ap.add_argument("--exchange", default="native", choices=["native", "torch"],
                    help="N > 1: native = pt_exchange_merge_dev (RCCL behind the C ABI: one count-matrix all-gather, grouped send/recv) -- if its "
                         "communicator does not come up the run FAILS (non-zero exit), it never changes protocol by itself; "
                         "torch = the torch.distributed all-gather protocol of sharding.py (what the gloo CPU tests drive), an explicit choice")
    ap.add_argument("--attributes", default="auto", choices=["auto", "sharded", "replicated"],
                    help="N > 1 with the native exchange: sharded = every rank generates its slab in index order, keeps positions in its records "
                         "and the attribute records of its OWN points only (16 n / N bytes; the answers carry their candidates' records; the lists "
                         "are translated to global indices by one more pass, ~ 1 ms per 50 M neighbours); replicated = every rank holds the whole "
                         "16 n-byte table (what --exchange torch always does); auto = replicated while the whole table is under a quarter of the "
                         "GPU's memory (config 4: 16 GB of 288), sharded beyond")
    ap.add_argument("--source-points", dest="n", type=int, default=0, help="override the workload's source count (rehearsals)")
    ap.add_argument("--target-points", dest="m", type=int, default=0, help="override the workload's target count (rehearsals)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch + rendezvous + fence + max-over-ranks timing only, no GPU work: what the CPU suite drives (gloo). The JSON "
                         "line says dry_run and carries no value")
    ap.add_argument("--fail-rank", type=int, default=-1, help="rehearsal of a failing rank: that rank exits with code 9 before the rendezvous")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0:
        if args.gpus > 1:           # plain start: become the launcher -- decided before torch is imported or any GPU call is made
            sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
        world = 1
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but the launcher's WORLD_SIZE is %d" % (args.gpus, world))
    if rank == args.fail_rank:
        sys.exit(9)

    import torch
    import torch.distributed as dist
    if args.dry_run:
        return dry_run(args, world, rank, dist, torch)
    # device_count() does not initialise the GPU on this image: refuse BEFORE any rank enters a collective the others would wait in
    if args.backend == "nccl" and torch.cuda.device_count() < world:
        sys.exit("bench.py: --gpus %d needs %d visible GPUs, this machine shows %d" % (world, world, torch.cuda.device_count()))
    import __graft_entry__ as g
    pkg = g.load_package()
    from pt_amd import sharding

    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
            comm = sharding.HostStagedComm()
        else:
            dist.init_process_group("nccl", device_id=dev)
            comm = sharding.TorchDistComm()
    else:
        comm = sharding.SingleComm()

    n_total, m_total, k, seed = WORKLOADS[args.workload]
    if args.n:
        n_total = args.n
    if args.m:
        m_total = args.m
    axis = 2                   # slabs are cut along z: the macro blocks are laid out x-fastest, and a slab that keeps whole x / y rows of them searches 8 - 10 % faster than one cut along x
                               # (tools/rehearse_slabs_c4.py 8 1e9 0 C4 0|2: 3.38 vs 3.06 ms per slab at G = 8; config 5: 27.4 vs 25.2)
    bounds = sharding.uniform_slab_bounds(world)
    pt = pkg.PointsTransfer(device=local_rank, k_hint=k)
    native = world > 1 and args.exchange == "native"
    if native and args.backend != "nccl":
        sys.exit("bench.py: --exchange native needs --backend nccl (RCCL places one rank per GPU); the gloo rehearsal takes --exchange torch")
    if native:
        # The library's own RCCL communicator, joined BEFORE the heavy GPU work (include/pt_api.h); rank 0 creates the id and
        # torch.distributed only carries its 128 bytes.  Any failure here ends the rank with a non-zero code -- the launcher then stops
        # the others, which may already be inside the collective ncclCommInitRank -- and NEVER switches to another protocol: a run that
        # was asked for the native exchange measures the native exchange or nothing.
        try:
            uid = [pt.comm_unique_id() if rank == 0 else None]
        except Exception as e:                      # noqa: BLE001
            print("bench.py: rank %d: pt_comm_unique_id failed (%s): is librccl loadable?" % (rank, e), file=sys.stderr)
            uid = [None]
        dist.broadcast_object_list(uid, src=0, device=dev)
        if uid[0] is None:
            sys.exit("bench.py: rank %d: no RCCL id from rank 0 -- the native exchange cannot start (use --exchange torch to measure the other protocol)" % rank)
        try:
            pt.comm_init(world, rank, uid[0])
        except Exception as e:                      # noqa: BLE001
            sys.exit("bench.py: rank %d: pt_comm_init failed (%s)" % (rank, e))
    dist_name, type_name = WORKLOAD_KIND.get(args.workload, ("uniform", "f32"))
    gen = dict(dist=pkg.capi.DIST_CLUSTERED if dist_name == "clustered" else pkg.capi.DIST_UNIFORM,
               xyz_type=pkg.F16 if type_name == "f16" else pkg.F32)
    if dist_name == "clustered" and world > 1:
        # equal-count slabs of a non-uniform cloud: quantiles of a sample of the generator's own output.  The generator is
        # index-addressable and i.i.d., so every rank derives the same bounds from the same small prefix: no communication.
        with pkg.PointsTransfer(device=local_rank) as probe:
            probe.build_synth(4_000_000, seed, **gen)
            probe.targets_synth(1_000_000, seed, **gen)
            sx = torch.empty((3, probe.num_targets), dtype=torch.float32, device=dev)
            probe.resident_target_xyz_dev(sx)
            bounds = sharding.quantile_slab_bounds(sx[axis], world)
    hbm = torch.cuda.get_device_properties(local_rank).total_memory
    sharded_attr = native and (args.attributes == "sharded" or (args.attributes == "auto" and 16 * n_total > hbm // 4))
    if sharded_attr:
        pt.set_param("local_ids", 1)
    if world > 1:
        pt.build_synth(n_total, seed, slab_axis=axis, slab_lo=bounds[rank], slab_hi=bounds[rank + 1], **gen)
        pt.targets_synth(m_total, seed, slab_axis=axis, slab_lo=bounds[rank], slab_hi=bounds[rank + 1], **gen)
    else:
        pt.build_synth(n_total, seed, **gen)
        pt.targets_synth(m_total, seed, **gen)
    n_loc, m_loc = pt.num_source, pt.num_targets
    st_first = pt.stats()          # the FIRST build of this cloud (allocation of every buffer included): what a one-shot run pays
    idx = torch.empty((m_loc, k), dtype=torch.int32, device=dev)
    d2 = torch.empty((m_loc, k), dtype=torch.float64, device=dev)
    rgb = torch.empty((m_loc, 3), dtype=torch.float32, device=dev)
    nrm = torch.empty((m_loc, 3), dtype=torch.float32, device=dev)
    xyz = torch.empty((3, m_loc), dtype=torch.float32, device=dev)
    pt.resident_target_xyz_dev(xyz)
    engine = sharding.GpuSlabEngine(pt, pkg.F32, dev)
    with_pca = args.workload == "C3" and world == 1          # PCA needs the whole cloud resident (no slabs)
    pnrm = torch.empty((m_loc, 3), dtype=torch.float32, device=dev) if with_pca else None

    kms = [0.0] * 8
    phase = {"build": 0.0, "target_sort": 0.0, "knn_blend_fused": 0.0, "reblend_merged": 0.0}
    xstats = {}
    flags = {}

    def reblend(rows):
        # targets whose neighbour lists were completed by other slabs: their fused blend is redone from the merged lists
        c = int(rows.numel())
        if not c:
            return
        r2 = torch.empty((c, 3), dtype=torch.float32, device=dev); n2 = torch.empty((c, 3), dtype=torch.float32, device=dev)
        pt.blend_dev(idx[rows].contiguous(), d2[rows].contiguous(), c, k, pkg.BLEND_MEAN, r2, n2)
        rgb[rows] = r2; nrm[rows] = n2

    def step(record):
        pt.rebuild()
        # k-NN and blend in one pass: the tile kernel gathers the attribute records as it settles each target
        pt.query_blend_resident_dev(k, pkg.BLEND_MEAN, idx, d2, rgb, nrm)
        st = pt.stats() if record else None       # HIP-event times of the build + home search, on the stream they ran on
        if world > 1 and record:                   # (the exchange synchronises with the host anyway: its wall time is a fair phase time)
            torch.cuda.synchronize(); tx = time.perf_counter()
        if native:         # count matrix -> owner-to-owner requests -> bounded answers -> merge -> re-blend, all behind the C ABI
            xs = pt.exchange_merge_dev(xyz, pkg.F32, m_loc, k, axis, bounds, idx, d2, pkg.BLEND_MEAN, rgb, nrm, want_stats=record)
        else:
            xs = sharding.exchange_and_merge(comm, engine, xyz, idx, d2, k, axis, bounds, on_changed=reblend)
        if world > 1 and record:
            torch.cuda.synchronize(); phase["exchange_merge_reblend"] = phase.get("exchange_merge_reblend", 0.0) + (time.perf_counter() - tx) * 1e3
        if with_pca:
            pt.pca_normals_dev(idx, m_loc, k, pnrm)     # BASELINE config 3: PCA normal estimation from the neighbours
        if record:
            if with_pca:
                phase["pca"] = phase.get("pca", 0.0) + pt.stats()["ms_pca"]
            for i in range(8):
                kms[i] += st["ms_kernel"][i]
            phase["build"] += st["ms_build"]; phase["target_sort"] += st["ms_sort_targets"]
            phase["knn_blend_fused"] += st["ms_query"]; phase["reblend_merged"] += pt.stats()["ms_blend"]
            xstats.update(xs)
            xstats["tile_leftover"] = st["n_leftover"]
            flags.update({f: st[f] for f in ("pass1_pooled", "pass2_pooled", "uniform_probe", "bbox_guess")})

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    box = BoxSampler(torch, local_rank)
    # the first step after the first build: what a run that builds ONCE (reference src/pointsTransfer.cpp:259) pays for its search
    torch.cuda.synchronize(); tq = time.perf_counter()
    pt.query_blend_resident_dev(k, pkg.BLEND_MEAN, idx, d2, rgb, nrm)
    torch.cuda.synchronize(); first_query_wall = (time.perf_counter() - tq) * 1e3
    st_q = pt.stats()
    first = {"first_build_ms": st_first["ms_build"], "first_build_kernels_ms": [round(v, 4) for v in st_first["ms_kernel"][:6]],
             "first_query_ms": st_q["ms_sort_targets"] + st_q["ms_query"], "first_query_wall_ms": first_query_wall,
             "first_build_flags": {f: st_first[f] for f in ("pass1_pooled", "pass2_pooled", "uniform_probe", "bbox_guess", "n_refine", "presort_refine", "n_sorts")}}
    for _ in range(args.warmup):
        step(False)
    fence()
    box.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    box.stop()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = m_total * args.steps / dt
    # COLD steps (one GPU): every build forgets what earlier builds of this resident cloud learnt ("forget": sampled bounding box, pooled
    # passes, uniformity sample, cell size are decided afresh, as on a first build) -- buffers stay allocated, so this is the first build's
    # device work without hipMalloc.  `value` stays the warm figure (the contract's K timed steps); both are printed.
    cold = None
    if world == 1:
        cs, cb = [], []
        for _ in range(3):
            torch.cuda.synchronize(); tc = time.perf_counter()
            pt.set_param("forget", 1)
            pt.rebuild()
            pt.query_blend_resident_dev(k, pkg.BLEND_MEAN, idx, d2, rgb, nrm)
            torch.cuda.synchronize(); cs.append((time.perf_counter() - tc) * 1e3)
            stc = pt.stats(); cb.append(stc["ms_build"])
        cold = dict(first, cold_step_ms=round(sum(cs) / len(cs), 4), cold_build_ms=round(sum(cb) / len(cb), 4),
                    cold_flags={f: stc[f] for f in ("pass1_pooled", "pass2_pooled", "uniform_probe", "bbox_guess", "n_refine", "presort_refine", "n_sorts")},
                    note="cold_step = forget + rebuild + query on allocated buffers (wall, mean of 3); first_build_ms includes every hipMalloc of the first build")

    if rank == 0:
        K = args.steps
        kavg = [v / K for v in kms]
        dom = max(range(8), key=lambda i: kavg[i])
        sx = 6 if type_name == "f16" else 12                                  # s = bytes per xyz (fp16 clouds: 6)
        p1, p2 = flags.get("pass1_pooled", 0) == 1, flags.get("pass2_pooled", 0) == 1      # which passes the timed steps ran (the byte model follows them)
        alg = survey_alg_bytes(KERNEL_NAMES[dom], n_loc, m_loc, k, sx, p1, p2)      # what frac is quoted on
        impl = kernel_alg_bytes(KERNEL_NAMES[dom], n_loc, m_loc, k, 12, p1, p2)     # this build's own minimum (its sorted records are fp32 either way), for comparison only
        achieved = alg / (kavg[dom] * 1e-3) / 1e9
        _, pmc_file = pmc_profile(args.workload, world)
        b_alg_job = n_total * (2 * sx + 4) + (n_total * sx + m_total * sx + m_total * k * 16 + m_total * (4 * k + 24))   # B_build + B_query
        out = {
            "metric": "target points/sec (k=%d detail transfer)" % k,
            "value": value, "unit": "target points/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d-point source / %d targets / k=%d, %s %s xyz in the unit cube, generator seed 0x%X"
                                   % (args.workload, n_total, m_total, k, dist_name, type_name, seed),
                       "step": "grid build + target binning + k-NN with fused mean blend + slab exchange/merge%s, inputs resident in HBM"
                               % (" + PCA normals" if with_pca else ""),
                       "parallelism": "slab%d" % world if world > 1 else "single", "backend": args.backend if world > 1 else None,
                       "exchange": ("native RCCL (pt_exchange_merge_dev)" if native else "torch.distributed all-gather (--exchange torch)") if world > 1 else None,
                       "attributes": ("sharded: 16 n / N bytes per rank" if sharded_attr else "replicated: 16 n bytes per rank") if world > 1 else None},
            "roofline": {"bound": "hbm", "kernel": KERNEL_NAMES[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(KERNEL_NAMES[dom], args.workload, world),
                         "traffic_source": pmc_file, "alg_bytes_per_launch": alg, "alg_bytes_formula": "SURVEY.md 8(d)",
                         "impl_bytes_per_launch": impl, "impl_achieved": impl / (kavg[dom] * 1e-3) / 1e9, "avg_launch_ms": kavg[dom]},
            "job_roofline": {"alg_bytes_per_step": b_alg_job, "achieved": b_alg_job / (ms_per_step * 1e-3) / 1e9 / world, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s per GPU", "frac": b_alg_job / (ms_per_step * 1e-3) / 1e9 / world / HBM_PEAK_GBS,
                             # HBM bytes every kernel of one step actually moved (same PMC passes, summed over the step's dispatches)
                             "traffic_per_step": pmc_traffic("__step__", args.workload, world)},
            "kernels_ms": dict(zip(KERNEL_NAMES, [round(v, 4) for v in kavg])),
            "build_flags": flags,      # of the timed (warm) steps: 1 = that pass ran without its histogram; uniform_probe 0 = a previous build already knew
            "cold": cold,
            "phases_ms": {a: round(b / K, 4) for a, b in phase.items()},
            "rank0": {"n_source": n_loc, "n_target": m_loc, "exchange": xstats},
            # the state of rank 0's GPU over the timed steps (sysfs, sampled by a host thread) and what the pass-1 scatter -- the kernel
            # whose time moves most from box to box -- reached on it: 12 B read + 16 B record + 2 B id written per source point
            "box": dict(box.summary(), pass1_scatter_gbs=round(n_loc * 30 / (kavg[2] * 1e-3) / 1e9, 1) if kavg[2] > 0 else None),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n_total, m_total, k, seed)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if native:
        pt.comm_destroy()
    pt.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
