import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscenedetect_amd import engine as E
eng = E.ScoringEngine(0)
N, H, W = int(os.environ.get("ET_N", 64)), 1080, 1920
x = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device="cuda")
if os.environ.get("ET_SMOOTH"):
    base = torch.nn.functional.interpolate(torch.rand((1, 3, 9, 16), device="cuda") * 255, size=(H, W), mode="bilinear", align_corners=True)[0].permute(1, 2, 0)
    for i in range(N):
        x[i] = (base + torch.randn((H, W, 3), device="cuda") * 2).round().clamp(0, 255).to(torch.uint8)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); eng.score_device(x.data_ptr(), N, H, W, flags=E.SCORE_EDGES); ts.append(time.perf_counter() - t0)
dt = min(ts[1:])
print(f"edges N={N}: {dt*1e3:.2f} ms  {N/dt:.0f} fps   (calls 2-5: {' '.join('%.2f' % (t * 1e3) for t in ts[1:])} ms)")
