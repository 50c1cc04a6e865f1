"""Timeline of the LAST step out of a rocprofv3 --kernel-trace CSV: every dispatch's start (us after the step's first kernel), duration and the
idle gap in front of it -- where a small configuration's step goes that its kernels' durations do not explain (dev tool).
usage: python tools/timeline.py <..._kernel_trace.csv> [first-kernel-substring, default hist/scatter of the rebuild]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "bbox"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# the last step starts at the last dispatch of `first` that is preceded by a different kernel
cand = [i for i in starts if i == 0 or first not in rows[i - 1]["Kernel_Name"]]
i0 = cand[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
busy = 0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
    print("%9.1f us  dur %8.1f  gap %7.1f  %s  grid %s wg %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
    busy += e - s; prev_end = max(prev_end, e)
print("step: %.1f us from first start to last end, %.1f us inside kernels, %d dispatches" % ((prev_end - t0) / 1e3, busy / 1e3, len(rows) - i0))
