"""Edge term + HSV on S / T / U content with whatever library PSD_LIB_PATH names (one process per library: run it
alternately for an A/B on one box).  Prints the HIP-event time of the whole term per submission and the step time with two
submissions in flight (bench.py's step), and a checksum of the records so that two builds can be told identical.
usage: python tools/edge_ab.py [frames=2048] [dists=STU] [tag]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from pyscenedetect_amd import engine as E, epilogue

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dists = sys.argv[2] if len(sys.argv) > 2 else "STU"
tag = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(os.environ.get("PSD_LIB_PATH", "default"))
dev = torch.device("cuda", 0)
eng = E.ScoringEngine(0)
for d in dists:
    batch = bench.make_batch(n, d, 20250921, dev, 1080, 1920)
    wl = bench.Workload(eng, batch, "edges", None, epilogue, E)
    wl.submit(); wl.submit(); wl.finish(); wl.finish()
    torch.cuda.synchronize()
    ms = []
    t0 = time.perf_counter()
    wl.submit()
    for _ in range(7):
        wl.submit(); ms.append(wl.finish())
    ms.append(wl.finish())
    dt = (time.perf_counter() - t0) / 8
    recs = wl.state["recs"]
    crc = zlib.crc32(np.stack([recs[f] for f in ("sad_h", "sad_s", "sad_v", "edge_xor")]).tobytes())
    px = n * 1080 * 1920
    print(f"{tag:>12} {d} n={n}: kernels {np.mean(ms):7.3f} ms  step {dt*1e3:7.3f} ms  {n/dt/1e3:7.1f} k frames/s  "
          f"of 5 B/px at 8 TB/s: {px*5/(np.mean(ms)*1e-3)/8e12:.4f}  records crc {crc:08x}", flush=True)
    del wl, batch
    torch.cuda.empty_cache()
