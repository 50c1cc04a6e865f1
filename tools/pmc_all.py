"""rocprofv3 --pmc CSV: every dispatch of the kernels whose name contains argv[2] (dev helper): python tools/pmc_all.py file.csv knn_wave"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for did, c in agg.items():
    print(did, "  ".join("%s=%.4g" % kv for kv in sorted(c.items())))
