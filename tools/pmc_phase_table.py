"""Summary of tools/pmc_phases.sh: per build (staging / + pass 1 / + pass 2 / full / full + blend) the tile kernel's counters of its LAST
dispatch, and the per-phase differences.  usage: python tools/pmc_phase_table.py gpurun_out/r03/phases"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
labels = ["stage", "pass1", "pass2", "full", "fullblend"]
data = collections.OrderedDict()
for lab in labels:
    vals = {}
    for f in glob.glob(os.path.join(root, lab + "_*", "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "knn_tile_kernel" in r["Kernel_Name"]]
        if not rows:
            continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ts = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if int(r["Dispatch_Id"]) == last]
        vals["ms(" + os.path.basename(os.path.dirname(os.path.dirname(f))).split("_")[-1] + ")"] = ts[0] / 1e6 if ts else 0.0
    data[lab] = vals
keys = sorted({k for v in data.values() for k in v})
print("%-22s" % "counter (1e9 / launch)" + "".join("%12s" % l for l in labels) + "   | per phase: " + " ".join(["staging", "pass1", "pass2", "pass3", "blend"]))
for k in keys:
    row = [data[l].get(k) for l in labels]
    scale = 1.0 if k.startswith("ms(") else 1e-9
    cells = "".join("%12s" % ("%.3f" % (v * scale) if v is not None else "-") for v in row)
    diffs = []
    prev = 0.0
    for v in row:
        diffs.append("%.3f" % ((v - prev) * scale) if v is not None else "-")
        prev = v if v is not None else prev
    print("%-22s" % k + cells + "   | " + " ".join(diffs))
