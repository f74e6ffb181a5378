"""round 5, probe 21: how long does the slow state after an idle GPU last?  Downscale workload (1.15 ms kernel), two steps in flight, wall time per
10 steps after a 2 s pause; same for the hash workload and all-four-downscale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pyscenedetect_amd import engine as E, epilogue
dev = torch.device("cuda", 0)
eng = E.ScoringEngine(0)
batch = bench.make_batch(4096, "U", 20250921, dev, 1080, 1920)
for det, ds in (("content", "auto"), ("all", "auto"), ("content", None)):
    wl = bench.Workload(eng, batch, det, ds, epilogue, E)
    wl.submit(); wl.finish(); torch.cuda.synchronize()
    time.sleep(2.0)
    out = []
    wl.submit()
    t0 = time.perf_counter()
    for i in range(1, 161 if ds else 41):
        wl.submit(); wl.finish()
        if i % 10 == 0:
            t1 = time.perf_counter(); out.append(round((t1 - t0) / 10 * 1e3, 3)); t0 = t1
    wl.finish(); torch.cuda.synchronize()
    print(det, ds, "ms per step by tens:", out, flush=True)
