"""Residency curve of one launch of the staged scoring kernel from a -DPSD_WG_TIMELINE=1 build (tools/ablate.sh -f tl ...):
every workgroup's start / end on the 100 MHz clock and the hardware id of its first wave.
usage: PSD_LIB_PATH=.../libpsd_tl.so python tools/wg_timeline.py [frames] [hsv|all]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscenedetect_amd import engine as E, _native

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
what = sys.argv[2] if len(sys.argv) > 2 else "hsv"
H, W = 1080, 1920
eng = E.ScoringEngine(0)
lib = _native.load()
lib.psd_debug_timeline.argtypes = [ctypes.c_void_p]
x = torch.empty((N, H, W, 3), dtype=torch.uint8, device="cuda")
g = torch.Generator(device="cuda").manual_seed(3)
for i in range(0, N, 64):
    x[i:i + 64] = torch.randint(0, 256, (min(64, N - i), H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
buf = torch.zeros((1 << 20, 3), dtype=torch.int64, device="cuda")       # up to 1 M workgroups
assert lib.psd_debug_timeline(buf.data_ptr()) == 0
flags = E.SCORE_HSV_SAD if what == "hsv" else 7
for _ in range(3):
    buf.zero_()
    torch.cuda.synchronize()
    eng.score_device(x.data_ptr(), N, H, W, flags=flags)
ms = eng.last_kernel_ms()[0]
t = buf.cpu().numpy()
t = t[t[:, 1] > 0]
t0, t1, hw = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64), t[:, 2]
base = t0.min()
s, e = (t0 - base) / 100.0, (t1 - base) / 100.0            # microseconds (100 MHz clock)
dur = e - s
print(f"{what} N={N}: kernel {ms:.3f} ms by HIP events, {len(t)} workgroups, span {e.max() / 1e3:.3f} ms; workgroup duration "
      f"mean {dur.mean():.1f} us, p5 {np.percentile(dur, 5):.1f}, p50 {np.percentile(dur, 50):.1f}, p95 {np.percentile(dur, 95):.1f}")
# residency over time: workgroups alive in 20 slices
edges = np.linspace(0, e.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2
    alive = np.count_nonzero((s <= mid) & (e > mid))
    started = np.count_nonzero((s >= a) & (s < b))
    d_here = dur[(s >= a) & (s < b)]
    print(f"  t = {mid / 1e3:6.3f} ms   resident {alive:5d}   started {started:5d}   mean duration of those {d_here.mean() if len(d_here) else 0:7.1f} us")
order = np.argsort(s)
first = order[: min(1536, len(order))]
print(f"first {len(first)} workgroups: starts within {s[first].max():.1f} us, their durations mean {dur[first].mean():.1f} us (all: {dur.mean():.1f})")
last = np.argsort(e)[-64:]
print(f"last 64 to finish: started at {s[last].mean() / 1e3:.3f} ms on average, duration {dur[last].mean():.1f} us")
# time between the end of the last-but-1536th workgroup and the end of the launch = the tail
ends = np.sort(e)
print(f"tail: 95 % of the workgroups are done at {ends[int(0.95 * len(ends))] / 1e3:.3f} ms, 99 % at {ends[int(0.99 * len(ends))] / 1e3:.3f} ms, all at {ends[-1] / 1e3:.3f} ms")
xcc = (hw >> 0) & 0xffffffff
print("distinct HW_ID values:", len(np.unique(xcc)))
