"""Shader clock and board power while an evaluation loop runs (sysfs / hwmon, sampled from a second
thread): is the evaluation clock- or power-limited?   usage: python3 tools/clock_probe.py [config] [nobs] [evaluations]"""
import glob, os, sys, threading, time
sys.path.insert(0, '.')
import numpy as np
from gogp_amd import configs, gp as G

def read(path):
    try:
        return open(path).read()
    except Exception as e:  # noqa: BLE001
        return "ERR %r" % (e,)

import ctypes, torch
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
buf = ctypes.create_string_buffer(64)
hip.hipDeviceGetPCIBusId(buf, 64, 0)
bus = buf.value.decode().lower()
base = "/sys/bus/pci/devices/" + bus
print("HIP device 0 is PCI", bus, "exists:", os.path.isdir(base), flush=True)
cards = sorted(glob.glob(base + "/pp_dpm_sclk"))
hw = sorted(glob.glob(base + "/hwmon/hwmon*/power1_average")) + sorted(glob.glob(base + "/hwmon/hwmon*/power1_input"))
fq = sorted(glob.glob(base + "/hwmon/hwmon*/freq1_input"))
print("sclk files:", cards[:2], "power files:", hw[:2], "freq files:", fq[:2], flush=True)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = configs.workload(cfg, int(sys.argv[2]) if len(sys.argv) > 2 else None)
NEV = int(sys.argv[3]) if len(sys.argv) > 3 else 40
X, y = wl.inputs()
g = G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0, precision=32 if wl.dtype == "f32" else 64)
g.Observe(wl.log_theta(0)); g.Gradient()
samples, stop = [], False
def sampler():
    while not stop:
        t = time.perf_counter()
        f = read(fq[0]).strip() if fq else ""
        p = read(hw[0]).strip() if hw else ""
        s = [l for l in read(cards[0]).splitlines() if "*" in l] if cards else []
        samples.append((t, f, p, s[0] if s else ""))
        time.sleep(0.005)
th = threading.Thread(target=sampler); th.start()
time.sleep(0.3)
t0 = time.perf_counter()
for k in range(NEV):
    g.Observe(wl.log_theta(k)); g.Gradient()
t1 = time.perf_counter()
time.sleep(0.3)
stop = True; th.join()
print("%d evaluations of config %d at N = %d: %.2f ms each" % (NEV, cfg, wl.N, (t1 - t0) / NEV * 1e3))
idle = [s for s in samples if s[0] < t0 - 0.05]
busy = [s for s in samples if t0 + 0.5 < s[0] < t1]
def summ(ss, name):
    fs = [float(s[1]) / 1e6 for s in ss if s[1] and not s[1].startswith("ERR")]
    ps = [float(s[2]) / 1e6 for s in ss if s[2] and not s[2].startswith("ERR")]
    print(name, "samples", len(ss), "freq MHz min/mean/max", (min(fs), sum(fs) / len(fs), max(fs)) if fs else None,
          "power W min/mean/max", (min(ps), sum(ps) / len(ps), max(ps)) if ps else None, "sclk line:", ss[len(ss) // 2][3] if ss else None)
summ(idle, "idle:"); summ(busy, "busy:")
print("raw example:", samples[len(samples) // 2])
g.close()
