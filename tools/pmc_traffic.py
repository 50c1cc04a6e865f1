"""Derive profiles/*_pmc_traffic.json from two rocprofv3 counter_collection CSVs (FETCH_SIZE pass, WRITE_SIZE pass).

Per kernel the LARGEST dispatch is kept (the full-size launch); HBM bytes = FETCH_SIZE[KB]*1024*2 + WRITE_SIZE[KB]*1024
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
f = largest(sys.argv[1], "FETCH_SIZE"); w = largest(sys.argv[2], "WRITE_SIZE")
out = {"workload": sys.argv[3] if len(sys.argv) > 3 else "", "kernels": {}}
for k in f:
    rd = f[k] * 1024 * 2; wr = w.get(k, 0.0) * 1024
    out["kernels"][k] = {"FETCH_SIZE_KB": f[k], "WRITE_SIZE_KB": w.get(k, 0.0), "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr}
json.dump(out, sys.stdout, indent=1)
