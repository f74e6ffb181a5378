"""round 5, probe 20: the downscale workload's step is 1.36 ms inside bench.py's `secondary` and 1.195 ms in `bench.py --downscale auto` for the same
1.15 ms kernel.  What precedes it?  quick_measure of the downscale workload (a) first thing in the process, (b) after 25 headline steps, (c) again,
(d) after a pause of 2 s, (e) with the records of the previous workload still referenced or not."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pyscenedetect_amd import engine as E, epilogue
dev = torch.device("cuda", 0)
eng = E.ScoringEngine(0)
batch = bench.make_batch(4096, "U", 20250921, dev, 1080, 1920)
def q(tag, det="content", ds="auto", steps=20, warmup=5):
    r = bench.quick_measure(bench.Workload(eng, batch, det, ds, epilogue, E), steps=steps, warmup=warmup, parity_frames=0)
    print(tag, r["ms_per_step"], r["avg_launch_ms"], flush=True)
q("a first thing")
q("a again")
q("headline", "content", None, 20, 5)
q("b after headline")
q("c again")
time.sleep(2.0)
q("d after 2 s")
q("all4", "all", "auto")
q("hash", "hash", None, 3, 1)
import cProfile, pstats
wl = bench.Workload(eng, batch, "content", "auto", epilogue, E)
wl.submit(); wl.finish(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
wl.submit()
for _ in range(19):
    wl.submit(); wl.finish()
wl.finish(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
