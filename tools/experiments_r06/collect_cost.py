"""How long does collect() take once the records are on the host?  (6144 records with histograms = 6.5 MB from the page-locked mirror into a
fresh numpy array)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyscenedetect_amd import engine as E
eng = E.ScoringEngine(0)
n = 6144
x = torch.randint(0, 256, (n, 144, 256, 3), dtype=torch.uint8, device="cuda")
for flags, sums in ((7, False), (1, True), (7, True)):
    ts = []
    for _ in range(12):
        eng.submit_device(x.data_ptr(), n, 144, 256, flags=flags)
        time.sleep(0.02)                      # the submission is long done
        t0 = time.perf_counter(); r = eng.collect(n, sums_only=sums); ts.append(time.perf_counter() - t0)
    print("flags", flags, "sums_only", sums, "collect of finished work: min %.3f ms median %.3f ms" % (min(ts) * 1e3, sorted(ts)[6] * 1e3), r.nbytes)
a = np.empty(n * 1064, np.uint8); b = np.ones(n * 1064, np.uint8)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); c = np.empty(n * 1064, np.uint8); c[:] = b; ts.append(time.perf_counter() - t0)
print("numpy: fresh 6.5 MB array + copy: %.3f ms; into an existing one: " % (min(ts) * 1e3), end="")
ts = []
for _ in range(10):
    t0 = time.perf_counter(); a[:] = b; ts.append(time.perf_counter() - t0)
print("%.3f ms" % (min(ts) * 1e3))
