"""Where the time of one process_frame() call goes (1080p, pageable host frames): each native step on its own clock."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pyscenedetect_amd as psd
from pyscenedetect_amd import engine as E
from pyscenedetect_amd.timecode import FrameTimecode

eng = E.ScoringEngine(0)
H, W, N = 1080, 1920, 96
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
bufs = [eng.alloc(H * W * 3), eng.alloc(H * W * 3)]
def best(fn, reps=5):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
    return b / N * 1e6
out = {}
def up():
    for t in range(N): bufs[t & 1].upload(frames[t].reshape(-1))
out["upload_blocking_us"] = best(up)
def up_un():
    for t in range(N): bufs[t & 1].upload_unordered(frames[t].reshape(-1))
out["upload_unordered_us"] = best(up_un)
def sc():
    for t in range(N): eng.score_device(bufs[t & 1].ptr, 1, H, W, d_prev=bufs[(t & 1) ^ 1].ptr, flags=1)
out["score_device_n1_us"] = best(sc)
def sc_noprev():
    for t in range(N): eng.score_device(bufs[t & 1].ptr, 1, H, W, flags=1)
out["score_device_n1_noprev_us"] = best(sc_noprev)
def both():
    for t in range(N):
        bufs[t & 1].upload(frames[t].reshape(-1))
        eng.score_device(bufs[t & 1].ptr, 1, H, W, d_prev=bufs[(t & 1) ^ 1].ptr, flags=1)
out["upload_plus_score_us"] = best(both)
pin = eng.pinned_array((N, H, W, 3)); pin[:] = frames
def both_pinned():
    for t in range(N):
        bufs[t & 1].upload(pin[t].reshape(-1))
        eng.score_device(bufs[t & 1].ptr, 1, H, W, d_prev=bufs[(t & 1) ^ 1].ptr, flags=1)
out["upload_plus_score_pinned_us"] = best(both_pinned)
tcs = [FrameTimecode(i, 25.0) for i in range(N)]
def pf():
    det = psd.ContentDetector(engine=eng)
    for t in range(N): det.process_frame(tcs[t], frames[t])
out["process_frame_us"] = best(pf)
def pf_all():
    dets = [psd.ContentDetector(engine=eng), psd.HistogramDetector(engine=eng), psd.ThresholdDetector(engine=eng)]
    for t in range(N):
        for d in dets: d.process_frame(tcs[t], frames[t])
out["process_frame_three_detectors_us"] = best(pf_all, 3)
print({k: round(v, 1) for k, v in out.items()})
