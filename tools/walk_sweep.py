"""HSV pass kernel time (HIP events, psd_last_kernel_ms) against batch length, frame size and clip-start flags.
usage: python tools/walk_sweep.py [case ...]   case = HxWxN[:segs][:dist]   (dist U | K; default U)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pyscenedetect_amd import engine as E

eng = E.ScoringEngine(0)
cases = sys.argv[1:] or ["1080x1920x1024", "1080x1920x2048", "1080x1920x4096", "1080x1920x8192", "360x640x36864", "360x640x73535",
                         "360x640x73535:11"]
for case in cases:
    parts = case.split(":")
    h, w, n = (int(v) for v in parts[0].split("x"))
    segs = int(parts[1]) if len(parts) > 1 and parts[1] else 0
    dist = parts[2] if len(parts) > 2 else "U"
    x = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    step = max(1, (256 << 20) // (h * w * 3))
    for i in range(0, n, step):
        k = min(step, n - i)
        if dist == "K":
            x[i:i + k] = 77
        else:
            x[i:i + k] = torch.randint(0, 256, (k, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    torch.cuda.synchronize()
    ms = []
    for _ in range(6):
        if segs:
            first = [i * (n // segs) for i in range(segs)]
            eng.score_device_segments(x.data_ptr(), n, h, w, first, flags=E.SCORE_HSV_SAD)
        else:
            eng.score_device(x.data_ptr(), n, h, w, flags=E.SCORE_HSV_SAD)
        ms.append(eng.last_kernel_ms()[0])
    best, med = min(ms[1:]), float(np.median(ms[1:]))
    gb = n * h * w * 3 / 1e9
    print(f"{case:28s} blocks={os.environ.get('PSD_HSV_BLOCKS', '-'):>6s}  kernel min {best:8.3f} ms  median {med:8.3f} ms   {gb / med:7.1f} GB/s  frac {gb / med / 8000:.4f}",
          flush=True)
    del x
    torch.cuda.empty_cache()
