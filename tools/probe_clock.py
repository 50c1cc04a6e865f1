import time, os, subprocess
t0 = time.time(); m0 = time.monotonic(); p0 = time.perf_counter()
print("nproc", os.cpu_count(), subprocess.getoutput("grep -m1 'model name' /proc/cpuinfo"), subprocess.getoutput("free -g | sed -n 2p"))
import torch
print("import torch wall", time.time() - t0)
x = torch.zeros(1 << 28, device="cuda")  # 1 GiB
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
t = time.time(); e0.record()
for _ in range(200): x.add_(1.0)
e1.record(); torch.cuda.synchronize()
print("200 x 2GiB rw: event ms", e0.elapsed_time(e1), "wall ms", (time.time() - t) * 1e3, "=> GB/s by event", 200 * 2 * 1.0737 / (e0.elapsed_time(e1) / 1e3))
t = time.time(); s = 0
for i in range(10_000_000): s += i
print("python 10M loop wall", time.time() - t, "monotonic total", time.monotonic() - m0, "perf", time.perf_counter() - p0, "date", subprocess.getoutput("date +%s.%N"))
