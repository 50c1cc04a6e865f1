"""Derive profiles/*_pmc_traffic.json from two rocprofv3 counter_collection CSVs (FETCH_SIZE pass, WRITE_SIZE pass).

Per kernel the largest dispatch OF THE LAST STEP is kept (the full-size launch of the timed path: round 3 took the largest of the whole
run, which for kernels the first build runs differently -- scatter_pool_kernel with block ids -- was not the step's; the run-wide
maximum stays beside it as `largest_in_run_hbm_bytes`); HBM bytes = FETCH_SIZE[KB]*1024*2 + WRITE_SIZE[KB]*1024
(MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads).
usage: pmc_traffic.py fetch.csv write.csv "<workload note>" > out.json
"""
import csv, json, re, sys
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)", n)
    return m.group(1) if m else n[:30]
def largest(path, counter):
    best = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"]); v = float(r["Counter_Value"])
        if v >= best.get(k, -1.0):
            best[k] = v
    return best
def per_dispatch(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[int(r["Dispatch_Id"])] = (short(r["Kernel_Name"]), float(r["Counter_Value"]))
    return out
def last_step(path, counter):
    """Sum over the dispatches of the LAST hot-path step of the run: from its (sampled) bbox kernel to the end."""
    d = per_dispatch(path, counter)
    ids = sorted(d)
    starts = [i for i, k in enumerate(ids) if d[k][0] in ("bbox_kernel", "bbox_sample_kernel")]
    return sum(d[k][1] for k in ids[starts[-1]:]) if starts else None
def largest_in_last_step(path, counter):
    d = per_dispatch(path, counter)
    ids = sorted(d)
    starts = [i for i, k in enumerate(ids) if d[k][0] in ("bbox_kernel", "bbox_sample_kernel")]
    best = {}
    for k in (ids[starts[-1]:] if starts else ids):
        name, v = d[k]
        if v >= best.get(name, -1.0):
            best[name] = v
    return best
fa, wa = largest(sys.argv[1], "FETCH_SIZE"), largest(sys.argv[2], "WRITE_SIZE")
f = largest_in_last_step(sys.argv[1], "FETCH_SIZE"); w = largest_in_last_step(sys.argv[2], "WRITE_SIZE")
for k in fa:                       # kernels that only run outside the step (generators, the first build's own) keep their run-wide figure
    if k not in f:
        f[k] = fa[k]; w.setdefault(k, wa.get(k, 0.0))
out = {"workload": sys.argv[3] if len(sys.argv) > 3 else "", "kernels": {}}
for k in f:
    rd = f[k] * 1024 * 2; wr = w.get(k, 0.0) * 1024
    out["kernels"][k] = {"FETCH_SIZE_KB": f[k], "WRITE_SIZE_KB": w.get(k, 0.0), "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                         "largest_in_run_hbm_bytes": fa.get(k, 0.0) * 2048 + wa.get(k, 0.0) * 1024}
sf, sw = last_step(sys.argv[1], "FETCH_SIZE"), last_step(sys.argv[2], "WRITE_SIZE")
if sf is not None and sw is not None:
    out["step"] = {"note": "all dispatches of the last step (rebuild .. blend)", "FETCH_SIZE_KB": sf, "WRITE_SIZE_KB": sw,
                   "hbm_read_bytes_corrected": sf * 2048, "hbm_write_bytes": sw * 1024, "hbm_bytes": sf * 2048 + sw * 1024}
json.dump(out, sys.stdout, indent=1)
