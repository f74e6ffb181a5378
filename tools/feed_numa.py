"""Host-feed rates by CPU affinity (NUMA node of the feeding threads).  usage: python tools/feed_numa.py [node|-1]"""
import json
import os
import sys
import time

node = int(sys.argv[1]) if len(sys.argv) > 1 else -1
if node >= 0:
    txt = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
    cpus = set()
    for part in txt.split(","):
        a, _, b = part.partition("-")
        cpus |= set(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, cpus)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import pyscenedetect_amd as psd  # noqa: E402
from pyscenedetect_amd import engine as E  # noqa: E402

eng = E.ScoringEngine(0)
H, W, N = 1080, 1920, 384
hf = np.random.default_rng(0).integers(0, 256, (N, H, W, 3), dtype=np.uint8)
rows = eng.downscale_source_rows(H, W, 144, 256, 1)
buf = eng.alloc(hf.nbytes)
views = [hf[t] for t in range(N)]


def best_of(fn, reps=4):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def up():
    for a in range(0, N, 16):
        buf.upload_rows_batch(views[a:a + 16], a * H * W * 3, rows, H * W * 3)
    eng.upload_fence(wait_on_host=True)


def sm(stats):
    m = psd.SceneManager(psd.StatsManager() if stats else None, engine=eng)
    m.add_detector(psd.ContentDetector(engine=eng))
    m.detect_scenes(psd.ArrayVideoStream(hf, 25.0))


gpu_node = None
try:
    import glob
    for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
        gpu_node = (gpu_node or []) + [int(open(p).read())]
except Exception:
    pass
out = {"affinity_node": node, "gpu_numa_nodes_sysfs": gpu_node, "cpus": len(os.sched_getaffinity(0)),
       "upload_batched_fps": round(N / best_of(up), 1), "scene_manager_plain_fps": round(N / best_of(lambda: sm(False)), 1),
       "scene_manager_stats_fps": round(N / best_of(lambda: sm(True)), 1),
       "whole_frames_GBps": round(hf.nbytes / best_of(lambda: buf.upload_unordered(hf.reshape(-1))) / 1e9, 2)}
print(json.dumps(out))
