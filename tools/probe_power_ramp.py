"""Socket power and shader clock of the GPU under the headline kernel, sample by sample (round 6, bench.py SmiPoll)."""
import sys, time, glob, threading
sys.path.insert(0, '.')
import torch
from pyscenedetect_amd import engine as E
# usage: python tools/probe_power_ramp.py [any]   -- socket power / shader clock every 0.1 s through 8 s of the headline kernel.
# By default the sysfs node of the device the kernel runs on (by PCI address); "any": the first hwmon node that answers -- which in a
# container may be ANOTHER tenant's GPU (round 6: 239 W / 105 MHz flat through the whole load).
pr = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
base = "/sys/class/drm/card*/device" if "any" in sys.argv[1:] else "/sys/bus/pci/devices/%s" % bdf
print("device", bdf, "nodes", glob.glob(base + "/hwmon/hwmon*/power1_input"))
pw = glob.glob(base + "/hwmon/hwmon*/power1_input")[0]
fq = glob.glob(base + "/hwmon/hwmon*/freq1_input")[0]
eng = E.ScoringEngine(0)
n, h, w = 4096, 1080, 1920
x = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
for a in range(0, n, 256):
    x[a:a+256] = torch.randint(0, 256, (256, h, w, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
time.sleep(2.0)
rows = []; stop = False
def poll():
    t0 = time.time()
    while not stop:
        rows.append((round(time.time() - t0, 2), int(open(pw).read()) // 1000000, int(open(fq).read()) // 1000000))
        time.sleep(0.1)
th = threading.Thread(target=poll); th.start()
time.sleep(0.5)
t0 = time.time()
eng.submit_device(x.data_ptr(), n, h, w, flags=1)
k = 0
while time.time() - t0 < 8.0:
    eng.submit_device(x.data_ptr(), n, h, w, flags=1); eng.collect(n, True); k += 1
eng.collect(n, True)
t_load = time.time() - t0
time.sleep(1.0)
stop = True; th.join()
print("steps", k, "load seconds", round(t_load, 2))
print(" ".join(f"{t}:{p}W/{f}MHz" for t, p, f in rows[::3]))
