"""Round 6: the all-four-detectors instance of the fused downscale kernel reads 1.19 or 1.32 ms per 4096 x 1080p by PROCESS
(profiles/r06_q_*).  One process: the kernel time of 60 back-to-back calls (min / median / max per third), the device addresses
of the batch, and the same for the HSV-only instance.  usage: python tools/experiments_r06/luma_bimodal_probe.py [frames]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np
import torch

from pyscenedetect_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sh, sw, dh, dw = 1080, 1920, 144, 256
x = torch.randint(0, 256, (n, sh, sw, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
eng = E.ScoringEngine(0)
out = {"frames": n, "batch_ptr": hex(x.data_ptr())}
for name, flags in (("all_four", 7), ("hsv", 1), ("luma_only", 6), ("all_four_again", 7)):
    ms = []
    for _ in range(60):
        eng.score_device_downscaled(x.data_ptr(), n, sh, sw, dh, dw, flags=flags)
        ms.append(eng.last_kernel_ms()[0])
    ms = np.array(ms)
    out[name] = {"first20": [round(float(v), 3) for v in (ms[:20].min(), np.median(ms[:20]), ms[:20].max())],
                 "last20": [round(float(v), 3) for v in (ms[40:].min(), np.median(ms[40:]), ms[40:].max())]}
print(json.dumps(out))
