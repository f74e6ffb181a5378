"""resize_walk_kernel<STORE>: kernel time of the plain resize of 4096 x 1080p -> 256 x 144 (PSD_RESIZE_STORE_VEC selects how pixels leave)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyscenedetect_amd import engine as E
n = 4096
x = torch.randint(0, 256, (n, 1080, 1920, 3), dtype=torch.uint8, device="cuda")
eng = E.ScoringEngine(0)
out = torch.empty((n, 144, 256, 3), dtype=torch.uint8, device="cuda")
import time
ms = []
for _ in range(12):
    eng.synchronize(); t0 = time.perf_counter()
    for _ in range(4): eng.resize_device(x.data_ptr(), n, 1080, 1920, out.data_ptr(), 144, 256)
    eng.synchronize(); ms.append((time.perf_counter() - t0) * 250)
hs = []
for _ in range(12):
    eng.score_device_downscaled(x.data_ptr(), n, 1080, 1920, 144, 256, flags=1)
    hs.append(eng.last_kernel_ms()[0])
print(json.dumps({"store_vec": os.environ.get("PSD_RESIZE_STORE_VEC", "1"), "resize_ms_min_med": [round(min(ms), 4), round(sorted(ms)[6], 4)], "hsv_ms_min_med": [round(min(hs), 4), round(sorted(hs)[6], 4)]}))
