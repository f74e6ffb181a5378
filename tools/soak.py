"""Soak test: the same batch scored many times (with unrelated GPU traffic on another stream) must
give bit-identical records every time, and the records must match the oracle on a sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyscenedetect_amd import engine as E
from oracle import lib as orc

eng = E.ScoringEngine(0)
N, H, W = 384, 1080, 1920
x = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device="cuda")
noise_a = torch.empty((256, 1024, 1024), dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
torch.cuda.synchronize()
bad = 0
for name, fl in (("hsv", 1), ("luma", 6), ("all", 7)):
    ref = eng.score_device(x.data_ptr(), N, H, W, flags=fl)
    want = orc.score_batch(x[:3].cpu().numpy(), flags=fl)
    for f in ("sad_h", "sad_s", "sad_v", "byte_sum", "hist"):
        assert np.array_equal(ref[f][:3], want[f]), (name, f)
    t0 = time.time(); reps = 0
    while time.time() - t0 < float(os.environ.get("SOAK_SECS", "20")):
        with torch.cuda.stream(side):
            noise_a.normal_()                  # unrelated traffic competing for CUs and HBM
        for _ in range(8):
            eng.submit_device(x.data_ptr(), N, H, W, flags=fl)
            eng.submit_device(x.data_ptr(), N, H, W, flags=fl)
            for _ in range(2):
                got = eng.collect(N)
                if got.tobytes() != ref.tobytes():
                    bad += 1
                reps += 1
    print(f"{name}: {reps} repetitions, mismatches so far {bad}")
# HashDetector thumbnails: same property
ref_t = eng.hash_thumbs_device(x.data_ptr(), N, H, W, 16)
assert np.array_equal(ref_t[:3], orc.hash_thumbs(x[:3].cpu().numpy(), 16))
t0 = time.time(); reps = 0
while time.time() - t0 < float(os.environ.get("SOAK_SECS", "20")):
    with torch.cuda.stream(side):
        noise_a.normal_()
    for _ in range(16):
        if eng.hash_thumbs_device(x.data_ptr(), N, H, W, 16).tobytes() != ref_t.tobytes():
            bad += 1
        reps += 1
print(f"hash thumbs: {reps} repetitions, mismatches so far {bad}")
print("SOAK", "CLEAN" if bad == 0 else f"FAILED ({bad})")
