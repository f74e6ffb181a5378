import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pyscenedetect_amd import engine as E
eng = E.ScoringEngine(0)
for (n, h, w) in ((4096, 144, 256), (64, 144, 256), (3636, 360, 640), (4096, 1080, 1920), (16384, 144, 256)):
    x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    for name, fl in (("hsv", 1), ("all", 7)):
        best = 1e9
        for _ in range(5):
            eng.score_device(x.data_ptr(), n, h, w, flags=fl); best = min(best, eng.last_kernel_ms()[0])
        print(f"{n}x{w}x{h} {name}: {best:.4f} ms  {n/best*1e3/1e6:.3f} Mfps  {n*h*w*3/best/1e6:.0f} GB/s")
    del x
