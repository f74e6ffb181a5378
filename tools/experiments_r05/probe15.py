"""round 5, probe 15: why does bench.py's per-frame line read 253 us per process_frame() call when tools/experiments_r05/per_frame_breakdown.py
reads 176 us on the same box?  Same loop over frames that live in (a) a numpy-allocated array (numpy asks for transparent huge pages for large
arrays), (b) the array torch's .cpu() returns (torch's CPU allocator does not), (c) = (b) after madvise(MADV_HUGEPAGE) + copy."""
import os, sys, time, ctypes, mmap
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import pyscenedetect_amd as psd
from pyscenedetect_amd import engine as E
from pyscenedetect_amd.timecode import FrameTimecode
dev = torch.device("cuda", 0)
eng = E.ScoringEngine(0)
N, H, W = 160, 1080, 1920
b = bench.make_batch(N, "S", 20250921, dev, H, W)
tcs = [FrameTimecode(i, 25.0) for i in range(N)]
def loop(frames):
    det = psd.ContentDetector(engine=eng)
    cuts = []
    for i in range(len(frames)): cuts += det.process_frame(tcs[i], frames[i])
    return cuts
def best(frames, reps=4):
    loop(frames[:8]); t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); loop(frames); t = min(t, time.perf_counter() - t0)
    return round(t / len(frames) * 1e6, 1)
def thp_of(a):
    # AnonHugePages of the mapping that holds a's data (kB), from /proc/self/smaps
    addr = a.__array_interface__["data"][0]; cur = None; out = None
    for line in open("/proc/self/smaps"):
        p = line.split()
        if "-" in p[0] and len(p) >= 5 and all(c in "0123456789abcdef-" for c in p[0]):
            lo, hi = (int(x, 16) for x in p[0].split("-")); cur = lo <= addr < hi
        elif cur and line.startswith("AnonHugePages:"): out = int(p[1])
    return out
res = {}
t_arr = b.cpu().numpy()
res["torch_cpu_us"] = best(t_arr); res["torch_cpu_thp_kB"] = thp_of(t_arr)
n_arr = np.empty((N, H, W, 3), np.uint8); torch.from_numpy(n_arr).copy_(b)
res["numpy_alloc_us"] = best(n_arr); res["numpy_alloc_thp_kB"] = thp_of(n_arr)
lst = [np.array(t_arr[i]) for i in range(N)]      # one numpy allocation per frame, as a decoder hands them over
res["numpy_per_frame_us"] = best(lst); res["numpy_per_frame_thp_kB"] = thp_of(lst[0])
print(res, open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
