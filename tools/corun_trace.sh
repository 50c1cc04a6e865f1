#!/bin/bash
# kernel trace of tools/probe_corun.py: do the tile kernel and the blend kernel overlap in time? (dev helper)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/corun_trace
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/corun_trace -- python3 tools/probe_corun.py > gpurun_out/corun_trace.log 2>&1
f=$(find gpurun_out/corun_trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows if "knn_tile" in r["Kernel_Name"] or "blend_kernel" in r["Kernel_Name"]]
ev.sort()
t0 = ev[0][0]
for s, e, nme in ev:
    print("%10.3f .. %10.3f ms  (%7.3f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, nme))
PY
