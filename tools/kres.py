"""Kernel resource usage of one .hip file (dev helper): python tools/kres.py pt_query.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3d-reconstruction-from-point-cloud_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + sys.argv[3:], cwd=d, capture_output=True, text=True).stderr
cur = None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")[:110]}
        continue
    if cur is None:
        if "error" in line: print(line)
        continue
    for key in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m: cur[key.split(" ")[0]] = int(m.group(1))
    if "LDS Size" in line:
        if flt in cur["name"]:
            print("%-112s vgpr %3d sgpr %3d scratch %4d occ %d lds %6d" % (cur["name"], cur.get("VGPRs", -1), cur.get("TotalSGPRs", -1), cur.get("ScratchSize", -1), cur.get("Occupancy", -1), cur.get("LDS", -1)))
        cur = None
