"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel (last dispatch of each name) counter values."""
import csv, sys, collections, re
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)", n)
    return m.group(1) if m else n[:30]
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        key = (short(r["Kernel_Name"]), r["Dispatch_Id"])
        agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    last = collections.OrderedDict()
    for (name, did), c in agg.items():
        last[name] = c            # keep the last dispatch of each kernel
    print("==", path)
    for name, c in last.items():
        print("%-24s" % name, "  ".join("%s=%.4g" % kv for kv in c.items()))
