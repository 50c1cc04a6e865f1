"""What this box's GPU is set to (dev probe; VERDICT r2 item 2: the same binary takes 49.5 ms on one MI355X and 54.2 on another).
Reads sysfs directly (no GPU initialisation) and asks rocm-smi / amd-smi for whatever they will tell an ordinary user.
    python tools/box_info.py [out.json]"""
import glob
import json
import os
import subprocess
import sys


def rd(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError as e:
        return "<%s>" % e.__class__.__name__


def sysfs():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        dev = os.path.join(card, "device")
        if not os.path.exists(os.path.join(dev, "pp_dpm_sclk")) and not os.path.exists(os.path.join(dev, "current_compute_partition")):
            continue
        c = {}
        for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "power_dpm_force_performance_level", "current_compute_partition",
                  "current_memory_partition", "available_compute_partition", "mem_info_vram_total", "mem_info_vram_used", "gpu_busy_percent",
                  "mem_busy_percent", "pp_power_profile_mode", "unique_id", "vbios_version", "xcp_config", "device", "revision"):
            p = os.path.join(dev, f)
            if os.path.exists(p):
                c[f] = rd(p)
        for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            for f in ("power1_cap", "power1_cap_max", "power1_cap_default", "power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input", "freq1_input", "freq2_input"):
                p = os.path.join(hw, f)
                if os.path.exists(p):
                    c["hwmon." + f] = rd(p)
        out[os.path.basename(card)] = c
    return out


def run(cmd):
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        return {"rc": r.returncode, "out": r.stdout[-6000:], "err": r.stderr[-1500:]}
    except Exception as e:      # noqa: BLE001
        return {"rc": -1, "err": repr(e)}


if __name__ == "__main__":
    info = {"sysfs": sysfs(), "cpu": rd("/proc/cpuinfo").split("model name")[1].split("\n")[0].strip(": \t") if "model name" in rd("/proc/cpuinfo") else "?",
            "nproc": os.cpu_count(), "kernel": rd("/proc/version")}
    for name, cmd in {"rocm-smi": ["rocm-smi", "--showclocks", "--showpower", "--showperflevel", "--showcomputepartition", "--showmemorypartition", "--showmaxpower", "--showtemp", "--json"],
                      "rocm-smi-a": ["rocm-smi", "-a"],
                      "amd-smi-metric": ["amd-smi", "metric", "--clock", "--power", "--temperature", "--json"],
                      "amd-smi-static": ["amd-smi", "static", "--limit", "--partition", "--vbios", "--asic", "--json"]}.items():
        info[name] = run(cmd)
    txt = json.dumps(info, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)
    print(txt[:12000])
