"""Where the time of the non-fused default pipeline goes: resize to 256 x 144 alone, scoring the small frames alone, both.
usage: python tools/downscale_split.py [frames]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscenedetect_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sh, sw, dh, dw = 1080, 1920, 144, 256
x = torch.randint(0, 256, (n, sh, sw, 3), dtype=torch.uint8, device="cuda")
small = torch.empty((n, dh, dw, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
eng = E.ScoringEngine(0)
def wall(f, reps=5):
    best = 1e9
    for _ in range(reps):
        eng.synchronize(); t0 = time.perf_counter(); f(); eng.synchronize(); best = min(best, time.perf_counter() - t0)
    return round(best * 1e3, 4)
ALL = E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM
LUMA = E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM
rows = {"frames": n}
rows["resize_store_ms"] = wall(lambda: eng.resize_device(x.data_ptr(), n, sh, sw, small.data_ptr(), dh, dw))
for name, fl in (("hsv", E.SCORE_HSV_SAD), ("luma", LUMA), ("all", ALL)):
    rows[f"score_small_{name}_ms"] = wall(lambda: eng.score_device(small.data_ptr(), n, dh, dw, flags=fl))
    rows[f"downscaled_{name}_ms"] = wall(lambda: eng.score_device_downscaled(x.data_ptr(), n, sh, sw, dh, dw, flags=fl))
    eng.score_device_downscaled(x.data_ptr(), n, sh, sw, dh, dw, flags=fl); rows[f"downscaled_{name}_kernel_ms"] = round(eng.last_kernel_ms()[0], 4)
print(json.dumps(rows))
