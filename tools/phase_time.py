"""Per-phase wave time of the time-walking scoring kernel (a -DPSD_PHASE_TIMING=1 build: tools/ablate.sh -f phases ...).
usage: PSD_LIB_PATH=.../libpsd_phases.so python tools/phase_time.py [hsv,all]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscenedetect_amd import engine as E, _native
N = int(os.environ.get("KT_N", "2048")); H, W = 1080, 1920
eng = E.ScoringEngine(0)
lib = _native.load()
lib.psd_debug_phases.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
x = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device="cuda")
dist = os.environ.get("KT_DIST", "U")
if dist == "K":
    x.copy_(torch.randint(0, 256, (N, 1, 1, 1), dtype=torch.uint8, device="cuda").expand_as(x))
torch.cuda.synchronize()
names = {"hsv": E.SCORE_HSV_SAD, "all": 7}
labels = ["dma wait", "stage->regs", "arith+drain", "barrier", "flush", "rest"]
for name in (sys.argv[1] if len(sys.argv) > 1 else "hsv,all").split(","):
    for _ in range(2):
        eng.score_device(x.data_ptr(), N, H, W, flags=names[name])
    out = (ctypes.c_ulonglong * 8)()
    lib.psd_debug_phases(out, 1)
    eng.score_device(x.data_ptr(), N, H, W, flags=names[name]); ms = eng.last_kernel_ms()[0]
    lib.psd_debug_phases(out, 1)
    tot = sum(out[i] for i in range(6))
    print(f"{name} dist={dist}: {ms:.3f} ms, {out[7]} waves, mean wave life {tot / max(out[7], 1):.0f} clocks, "
          f"shader clock seen by the waves {tot / max(out[6], 1) * 100:.0f} MHz")
    for i in range(6):
        print(f"   {labels[i]:12s} {100.0 * out[i] / tot:5.1f} %")
