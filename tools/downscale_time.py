"""Kernel time of the default pipeline (downscale to 256 x 144, then the HSV term) on resident 1080p frames.
usage: python tools/downscale_time.py [frames]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from pyscenedetect_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sh, sw, dh, dw = 1080, 1920, 144, 256
x = torch.randint(0, 256, (n, sh, sw, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
eng = E.ScoringEngine(0)
rows = {}
for name, flags in (("fused_hsv", E.SCORE_HSV_SAD), ("resize_then_all_terms", E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM)):
    best = 1e9
    for _ in range(5):
        eng.score_device_downscaled(x.data_ptr(), n, sh, sw, dh, dw, flags=flags)
        best = min(best, eng.last_kernel_ms()[0])
    touched = n * 2 * dh * sw * 3
    rows[name] = {"kernel_ms": round(best, 4), "frames_per_s": round(n / best * 1e3), "source_rows_GBps": round(touched / best / 1e6, 1),
                  "of_8TBps": round(touched / best / 1e6 / 8000, 4)}
full = eng.score_device(x.data_ptr(), n, sh, sw, flags=E.SCORE_HSV_SAD)
rows["full_resolution_hsv_ms"] = round(eng.last_kernel_ms()[0], 4)
print(json.dumps({"frames": n, "source": [sh, sw], "target": [dh, dw], **rows}))
