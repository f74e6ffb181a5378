"""Time the scoring kernel on a device-resident 1080p batch (no parity check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscenedetect_amd import engine as E
N = int(os.environ.get("KT_N", "1024")); H, W = 1080, 1920
eng = E.ScoringEngine(0)
x = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device="cuda")
if os.environ.get("KT_DIST") == "K":
    x.copy_(torch.randint(0, 256, (N, 1, 1, 1), dtype=torch.uint8, device="cuda").expand_as(x))
torch.cuda.synchronize()
names = {"hsv": E.SCORE_HSV_SAD, "luma": E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM, "all": 7}
for name in os.environ.get("KT_FLAGS", "hsv").split(","):
    best = 1e9
    for _ in range(6):
        eng.score_device(x.data_ptr(), N, H, W, flags=names[name]); best = min(best, eng.last_kernel_ms()[0])
    print(f"{os.environ.get('KT_TAG','')} {name}: {best:.3f} ms {N/best/1e3*1e3:.0f} fps {N*H*W*3/best/1e6:.0f} GB/s")
