"""GPU probe: parity of the HIP scoring path vs the oracle on small batches + first timings."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyscenedetect_amd import engine as E
from oracle import lib as O

eng = E.ScoringEngine(0)
rng = np.random.default_rng(1)
FL = E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM
ok = True
def cmp(a, b, tag):
    global ok
    good = True
    for f in ("sad_h", "sad_s", "sad_v", "byte_sum", "hist"):
        if not np.array_equal(a[f], b[f]):
            good = False
            bad = np.argwhere(a[f] != b[f])
            print("MISMATCH", tag, f, bad[:5].tolist(), a[f].reshape(len(a), -1)[bad[0][0]][:4], b[f].reshape(len(b), -1)[bad[0][0]][:4])
    print(("ok  " if good else "FAIL"), tag)
    ok &= good
for (n, h, w) in [(5, 36, 64), (7, 37, 53), (3, 144, 256), (4, 1, 1), (9, 16, 16), (33, 90, 160), (2, 1080, 1920)]:
    fr = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    ref = O.score_batch(fr)
    cmp(eng.score_host(fr, flags=FL), ref, f"host all {n}x{h}x{w}")
    pv = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    cmp(eng.score_host(fr, prev=pv, flags=FL), O.score_batch(fr, pv), f"host prev {n}x{h}x{w}")
    # device path, packed (unaligned when h*w*3 % 16 != 0 -> generic kernel)
    buf = eng.alloc(fr.nbytes + 64)
    buf.upload(fr.reshape(-1))
    cmp(eng.score_device(buf.ptr, n, h, w, flags=FL), ref, f"dev packed {n}x{h}x{w}")
    # flag subsets
    r = eng.score_device(buf.ptr, n, h, w, flags=E.SCORE_HSV_SAD)
    g = all(np.array_equal(r[f], ref[f]) for f in ("sad_h", "sad_s", "sad_v")) and not r["hist"].any() and not r["byte_sum"].any()
    print(("ok  " if g else "FAIL"), "hsv-only"); ok &= g
    r = eng.score_device(buf.ptr, n, h, w, flags=E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM)
    g = all(np.array_equal(r[f], ref[f]) for f in ("hist", "byte_sum")) and not r["sad_h"].any()
    print(("ok  " if g else "FAIL"), "luma-only"); ok &= g
    buf.free()
# constant frames (worst-case histogram contention) and a smooth ramp
for val in (0, 255, 128):
    fr = np.full((3, 72, 128, 3), val, np.uint8)
    cmp(eng.score_host(fr, flags=FL), O.score_batch(fr), f"const {val}")
print("PARITY", "GREEN" if ok else "RED")

# ---- timing on device-resident batches -------------------------------------------------------
import torch
H, W = 1080, 1920
for N in (int(os.environ.get('PROBE_N', '1024')),):
    x = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for name, fl in (("hsv", E.SCORE_HSV_SAD), ("luma", E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM), ("all", FL)):
        best = 1e9
        for it in range(6):
            t0 = time.perf_counter()
            r = eng.score_device(x.data_ptr(), N, H, W, flags=fl)
            wall = (time.perf_counter() - t0) * 1e3
            ms, nl = eng.last_kernel_ms()
            best = min(best, ms)
        gbs = N * H * W * 3 / (best * 1e-3) / 1e9
        print(f"N={N} {name}: kernel {best:.3f} ms  wall {wall:.3f} ms  {N/(best*1e-3):.0f} fps  {gbs:.0f} GB/s  ({gbs/8000*100:.1f}% of 8TB/s) launches={nl}")
    if N == 64:
        ref = O.score_batch(x[:8].cpu().numpy())
        r = eng.score_device(x.data_ptr(), 8, H, W, flags=FL)
        cmp(r, ref, "1080p dev 8 frames")
    del x
print("DONE", "GREEN" if ok else "RED")
