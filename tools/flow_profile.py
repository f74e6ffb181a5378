"""cProfile of one pass of a flow workload (bench.py --workload bbc|corpus at the secondary sizes) on the GPU box:
where the host time of a step goes next to its kernels.  usage: python tools/flow_profile.py bbc|corpus [frames]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pyscenedetect_amd import engine as E  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "bbc"


class A:
    corpus_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    bbc_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    height = width = 0


eng = E.ScoringEngine(0)
fw = bench.FlowWorkload(kind, eng, torch.device("cuda", 0), 0, 1, A, small=False)
for _ in range(3):
    fw.step()
ts, ks = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ks.append(fw.step())
    ts.append(time.perf_counter() - t0)
print(f"{kind}: step {min(ts) * 1e3:.3f} ms (median {sorted(ts)[len(ts) // 2] * 1e3:.3f}), kernels {min(ks):.3f} ms, frames {fw.frames_total}")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    fw.step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
