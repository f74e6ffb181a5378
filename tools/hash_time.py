"""Time the HashDetector thumbnail kernel (psd_hash_thumbs_device) on resident frames.
usage: python tools/hash_time.py [N [1080p|4k [size]]] ; prints one JSON line per configuration."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscenedetect_amd import engine as E

eng = E.ScoringEngine(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
only_res = sys.argv[2] if len(sys.argv) > 2 else None
only_size = int(sys.argv[3]) if len(sys.argv) > 3 else None
for (h, w, n) in ((1080, 1920, N), (2160, 3840, max(1, N // 4))):
    if only_res and only_res != {1080: "1080p", 2160: "4k"}[h]:
        continue
    x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for size in ((only_size,) if only_size else (16, 32)):
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            eng.hash_thumbs_device(x.data_ptr(), n, h, w, size)
            wall = time.perf_counter() - t0
            ms, _ = eng.last_kernel_ms()
            best = min(best, ms)
        fps = n / (best * 1e-3)
        print(json.dumps({"kernel": "gray_area_dma_kernel", "res": f"{w}x{h}", "n": n, "size": size, "kernel_ms": round(best, 4),
                          "fps": round(fps), "achieved_GBps": round(fps * h * w * 3 / 1e9, 1),
                          "frac_of_8TBps": round(fps * h * w * 3 / 8e12, 4), "wall_ms_last": round(wall * 1e3, 3)}))
    del x
