"""A/B of the edge term's lane forms inside ONE process (one box, one set of batches): PSD_EDGE_LANES / PSD_EDGE_CHUNKS are
read per call.  For S / T / U content: ms per submission (HIP events around the whole term) and per step (wall clock, two
steps in flight like bench.py), and a check that every form returns the records of the single-stream form bit for bit.
usage: python tools/edge_lanes_ab.py [frames=2048] [dists=STU]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from pyscenedetect_amd import engine as E, epilogue

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dists = sys.argv[2] if len(sys.argv) > 2 else "STU"
forms = [("1", "4"), ("2", "2"), ("2", "4"), ("2", "8"), ("1", "4")]
dev = torch.device("cuda", 0)
eng = E.ScoringEngine(0)
for d in dists:
    batch = bench.make_batch(n, d, 20250921, dev, 1080, 1920)
    wl = bench.Workload(eng, batch, "edges", None, epilogue, E)
    ref = None
    for lanes, chunks in forms:
        os.environ["PSD_EDGE_LANES"], os.environ["PSD_EDGE_CHUNKS"] = lanes, chunks
        wl.submit(); wl.submit(); wl.finish(); wl.finish()
        torch.cuda.synchronize()
        ms = []
        t0 = time.perf_counter()
        wl.submit()
        for _ in range(5):
            wl.submit(); ms.append(wl.finish())
        ms.append(wl.finish())
        dt = (time.perf_counter() - t0) / 6
        recs = wl.state["recs"]
        key = np.stack([recs[f] for f in ("sad_h", "sad_s", "sad_v", "edge_xor")])
        if ref is None:
            ref = key
        same = np.array_equal(ref, key)
        px = n * 1080 * 1920
        print(f"{d} n={n} lanes={lanes} chunks={chunks}: kernels {np.mean(ms):7.3f} ms  step {dt*1e3:7.3f} ms  {n/dt/1e3:7.1f} k frames/s  "
              f"of 5 B/px at 8 TB/s: {px*5/(np.mean(ms)*1e-3)/8e12:.4f}  records {'identical' if same else 'DIFFER'}", flush=True)
    del wl, batch
    torch.cuda.empty_cache()
